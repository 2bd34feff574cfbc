timeout 600 python -m pytest tests/test_gpu_edges.py -x -q 2>&1 | tail -8
