#!/bin/bash
# slam_local_ba_batch: per-kernel table of S windows (run on the GPU box: gpurun -- bash scripts/ba_batch_prof.sh [S] [tag])
S=${1:-128}
TAG=${2:-ba_batch}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 scripts/probes/ba_batch_time.py $S $3 > gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$TAG/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"].rjust(7))
PY
tail -3 gpurun_out/prof_$TAG.log
