python -m pytest tests/test_gpu_kpset.py tests/test_gpu_headline.py tests/test_gpu_device_frontend.py tests/test_gpu_lk.py -x -q 2>&1 | tail -4
python scripts/prof_headline.py 2>&1 | tail -1
