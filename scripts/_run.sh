python -m pytest tests/test_gpu_headline.py -x -q 2>&1 | tail -4
