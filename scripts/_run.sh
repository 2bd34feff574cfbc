# scratch: the command file handed to gpurun during development (overwritten freely)
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo rc $?
