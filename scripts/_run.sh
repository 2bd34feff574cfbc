timeout 600 python -m pytest tests/test_gpu_detect.py -q 2>&1 | grep -E "E  |FAILED|passed|failed" | head -8
