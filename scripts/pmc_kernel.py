"""Average PMC counter values per kernel name from a rocprofv3 counter_collection csv dir: python scripts/pmc_kernel.py DIR [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(k[0].ljust(28), k[1].ljust(28), f"{sum(v) / len(v):.4g}", len(v))
