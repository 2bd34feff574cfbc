python -m pytest tests/test_gpu_kpset.py -x -q 2>&1 | tail -30
