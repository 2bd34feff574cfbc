"""Per-(kernel, grid) min / median / max duration from a rocprofv3 --kernel-trace csv dir: python scripts/kernel_durations.py DIR"""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    acc[(r["Kernel_Name"].split("(")[0][-24:], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: -statistics.median(acc[k]) * len(acc[k])):
    v = sorted(acc[k])
    if statistics.median(v) > 20: print(k[0].ljust(24), str(k[1]).rjust(9), f"n={len(v):4d} min {v[0]:7.1f} med {statistics.median(v):7.1f} max {v[-1]:7.1f}")
