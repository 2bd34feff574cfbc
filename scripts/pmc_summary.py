import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(k[0][:24].ljust(24), k[1].rjust(9), k[2].ljust(12), round(sum(v) / len(v), 1), len(v))
