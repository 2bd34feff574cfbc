timeout 600 python bench.py --no-cpu > gpurun_out/b.json 2> gpurun_out/b.err; echo rc $?
