timeout 500 python bench.py --no-cpu --steps 100 --warmup 10 --no-sweep > gpurun_out/b1.json 2> gpurun_out/b1.err; echo "nosweep rc $?"; grep -c "capturing" gpurun_out/b1.err
timeout 500 python bench.py --no-cpu --steps 100 --warmup 10 > gpurun_out/b2.json 2> gpurun_out/b2.err; echo "sweep rc $?"; grep -c "capturing" gpurun_out/b2.err
timeout 500 python bench.py --no-cpu --steps 100 --warmup 10 --no-sweep --no-ba > gpurun_out/b3.json 2> gpurun_out/b3.err; echo "nosweep noba rc $?"; grep -c "capturing" gpurun_out/b3.err
