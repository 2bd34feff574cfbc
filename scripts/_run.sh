cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba -o ba -- python3 scripts/prof_ba.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba100 -o ba -- python3 scripts/prof_ba.py 100 40000 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, statistics
for d in ("prof_ba", "prof_ba100"):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-28:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(d)
    for k in sorted(acc, key=lambda k: -sum(acc[k]))[:9]:
        v = sorted(acc[k]); print(" ", k.ljust(28), f"n={len(v):4d} min {v[0]:7.1f} med {statistics.median(v):7.1f} max {v[-1]:7.1f} sum {sum(v):9.1f}")
PY
