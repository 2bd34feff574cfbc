timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --no-cpu > gpurun_out/b.json 2> gpurun_out/b.err; echo rc $?
