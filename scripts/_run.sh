for v in c d1 d2 c d1 d2; do
  echo "=== $v"
  SLAMHIP_LIB=$PWD/slam.jl_amd/libslamhip_lk_$v.so python scripts/prof_flow.py 32 2>&1 | tail -5 | head -2
done
