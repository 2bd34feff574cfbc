python -m pytest tests/test_gpu_configs.py tests/test_gpu_ba.py -x -q 2>&1 | tail -25
