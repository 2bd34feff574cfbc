# scratch: the command file handed to gpurun during development (overwritten freely)
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
