for i in 1 2; do
SLAMHIP_LIB=$PWD/scripts/ubench/libslamhip_base.so timeout 200 python scripts/prof_pose_5pt.py 2>&1 | tail -2 | sed 's/^/base /'
timeout 200 python scripts/prof_pose_5pt.py 2>&1 | tail -2 | sed 's/^/new  /'
done
timeout 600 python -m pytest tests/test_gpu_kpset.py tests/test_gpu_pose_batch.py tests/test_gpu_5pt.py tests/test_gpu_pose_fuzz.py -x -q 2>&1 | tail -1
