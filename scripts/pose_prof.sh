cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_pose
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pose -o pose -- python3 bench.py --only pose --no-cpu > gpurun_out/pose_profiled.json 2>/dev/null
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/prof_pose/pose_kernel_stats.csv")))[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"])
PY
tail -1 gpurun_out/pose_profiled.json | head -c 1500
