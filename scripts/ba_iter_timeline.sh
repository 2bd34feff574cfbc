#!/bin/bash
# kernel timeline of the LAST iterations of one single-window BA solve (P = 50): gpurun -- bash scripts/ba_iter_timeline.sh [P] [M]
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl_ba
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_ba -o tl -- python3 scripts/prof_ba.py ${1:-50} ${2:-10000} 2>/dev/null | tail -1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/tl_ba/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-44:-14]
t0 = int(rows[0]["Start_Timestamp"]); pe = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - pe) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f}  grid {r['Grid_Size_X']:>7} wg {r['Workgroup_Size_X']:>4}  {r['Kernel_Name'].split('(')[0][-30:]}")
    pe = e
PY
rm -rf gpurun_out/tl_ba
