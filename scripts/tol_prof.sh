#!/bin/bash
# tolerance-mode batch build: parity test, build time, per-kernel table (run on the GPU box: gpurun -- bash scripts/tol_prof.sh [S])
S=${1:-128}
timeout 300 python -m pytest tests/test_gpu_tol_batch.py -x -q 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python scripts/prof_pyr_batch.py $S 20 u8 tol
rm -rf gpurun_out/prof_tol
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tol -o tol -- python3 scripts/prof_pyr_batch.py $S 20 u8 tol > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_tol/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
