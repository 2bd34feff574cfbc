#!/bin/bash
# DESIGN 4 co-runner table: the headline loop (S = 128, bit-exact and tolerance mode) under several scheduling policies.
# run on the GPU box: gpurun -- bash scripts/corunner_table.sh
run() {
  env "$@" python bench.py --only headline,tolbatch --steps 12 --warmup 3 2>/dev/null | tail -1 > /dev/null
  python - "$*" <<PY
import json, sys
j = json.load(open("bench_detail.json"))
r = j["roofline"]; t = j["tolerance_mode"]["batch"]
print(f"{sys.argv[1]:48s} exact {j['value']:8.0f} f/s step {j['ms_per_step']:6.2f} ms build {r['avg_launch_us']/1e3:5.2f} ms (alone {r['isolated_launch_us']/1e3:5.2f}, x{r['avg_launch_us']/r['isolated_launch_us']:.2f}) | "
      f"tol {t['value']:8.0f} f/s build {t['roofline']['avg_launch_us']/1e3:5.2f} ms (alone {t['roofline']['isolated_launch_us']/1e3:5.2f}, x{t['roofline']['avg_launch_us']/t['roofline']['isolated_launch_us']:.2f})")
PY
}
run SLAM_BENCH_DUMMY=1
run SLAM_BENCH_TRACK_PRIO=0
run SLAM_BENCH_TRACK_PRIO=1
run SLAM_BENCH_SERIAL=1
run SLAM_BENCH_CU_SPLIT=32
run SLAM_BENCH_CU_SPLIT=64
run SLAM_BENCH_CU_SPLIT=96
run SLAM_BENCH_CU_SPLIT=64:stride
