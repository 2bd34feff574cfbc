for v in detold new detold new; do
  echo "== $v"
  if [ $v = new ]; then python scripts/prof_detect.py 32 2>&1 | tail -5; else SLAMHIP_LIB=$PWD/slam.jl_amd/libslamhip_detold.so python scripts/prof_detect.py 32 2>&1 | tail -5; fi
done
