python -m pytest tests/test_gpu_lk.py tests/test_gpu_kpset.py tests/test_gpu_batch.py tests/test_gpu_headline.py tests/test_gpu_edges.py tests/test_gpu_pyramid.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for v in gp g2 g3 gp g3; do
  echo "=== $v"
  SLAMHIP_LIB=$PWD/slam.jl_amd/libslamhip_$v.so python scripts/prof_flow.py 32 2>&1 | tail -5 | head -3
done
for v in gp g3 gp g3; do
  SLAMHIP_LIB=$PWD/slam.jl_amd/libslamhip_$v.so python scripts/prof_headline.py 2>&1 | tail -1
done
