timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_ba.py -x -q 2>&1 | tail -3
