SLAMHIP_LIB=$GRAFT_REPO_ROOT/slam.jl_amd/libslamhip_dett.so timeout 120 python scripts/prof_detect.py 32 2>&1 | grep "ncur 975" | tail -2
timeout 600 python -m pytest tests/test_gpu_detect.py tests/test_gpu_detect_batch.py tests/test_gpu_kpset.py -x -q 2>&1 | tail -3
timeout 120 python scripts/prof_detect.py 32 2>&1 | tail -5
