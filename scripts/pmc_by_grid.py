"""PMC counter averages per (kernel, grid size) from a rocprofv3 --pmc csv dir: python scripts/pmc_by_grid.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][-24:], int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        if any(t in k[0] for t in ("cols_fused","rows_ck","cum_fused")): print(k[0].ljust(24), str(k[1]).rjust(9), k[2].ljust(26), f"{sum(v) / len(v):.5g}", len(v))
