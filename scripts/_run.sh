# scratch: the command file handed to gpurun during development (overwritten freely)
python -m pytest tests -x -q -m gpu
