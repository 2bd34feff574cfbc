"""Per hardware queue: busy time, idle gaps and the kernels that follow the longest gaps, from a rocprofv3 --kernel-trace csv dir
(steady-state window of the last 60 ms): python scripts/queue_gaps.py DIR"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"]); w1 = t_end - 10_000_000; w0 = w1 - 60_000_000
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if w0 <= s <= w1: byq[r.get("Queue_Id", "?")].append((s, e, r["Kernel_Name"].split("(")[0][-32:]))
for q, ks in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ks); span = ks[-1][1] - ks[0][0]
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        gaps[(n0, n1)].append(max(0, s1 - e0) / 1e3)
    print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e6:.1f} ms of {span / 1e6:.1f} ms")
    for (n0, n1), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"    gap after {n0:32s} before {n1:32s} n={len(g):3d} mean {sum(g) / len(g):7.1f} us total {sum(g) / 1e3:6.2f} ms")
