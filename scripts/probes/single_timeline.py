"""Per-queue busy time and the kernel timeline of a few steady-state frames of the single-stream loop, from a rocprofv3 --kernel-trace csv of
scripts/prof_single_loop.py: python scripts/single_timeline.py DIR [window_us]"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 3000e3
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
w1 = t_end - 5_000_000; w0 = w1 - int(win)
sel = [r for r in rows if w0 <= int(r["Start_Timestamp"]) < w1]
busy = collections.defaultdict(int); cnt = collections.Counter(); kt = collections.defaultdict(int)
ev = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[r["Queue_Id"]] += e - s; cnt[r["Queue_Id"]] += 1; kt[r["Kernel_Name"].split("(")[0][-28:]] += e - s
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
# time with >= 1 / 2 / 3 / 4 kernels running
depth = 0; last = w0; at = collections.defaultdict(int)
for t, d in ev:
    at[depth] += t - last; last = t; depth += d
at[depth] += w1 - last
print(f"window {win / 1e3:.0f} us, {len(sel)} kernels")
print("busy per queue (us):", {q: round(v / 1e3) for q, v in sorted(busy.items())}, "launches:", dict(cnt))
print("time with k kernels running (us):", {k: round(v / 1e3) for k, v in sorted(at.items())})
print("kernel time by name (us):", {k: round(v / 1e3) for k, v in sorted(kt.items(), key=lambda x: -x[1])[:12]})
