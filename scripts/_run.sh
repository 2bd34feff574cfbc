timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -k "failed_factorisation" 2>&1 | tail -12
