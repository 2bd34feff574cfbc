"""Kernel timeline (start, duration, queue, stream) of a 12 ms steady-state window from a rocprofv3 --kernel-trace csv dir:
python scripts/kernel_timeline.py DIR"""
# Timeline around a key-frame step from a rocprofv3 kernel trace csv.
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# take a 12 ms window ending 20 ms before the end of the trace (steady state)
w1 = t_end - 20_000_000; w0 = w1 - 12_000_000
def short(n):
    n = n.split("(")[0]
    return n[-40:]
prev = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < w0 or s > w1: continue
    q = r.get("Queue_Id", "?"); st = r.get("Stream_Id", "?")
    print(f"{(s-w0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{q:>3} s{st:>3} {short(r['Kernel_Name'])}")
