"""timeline of the LAST burst of kernels in a rocprofv3 --kernel-trace csv (bursts are separated by > 2 ms of silence): python ... DIR"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
i0 = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 2_000_000:
        i0 = i
t0 = int(rows[i0]["Start_Timestamp"]); busy = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>2}  grid {r['Grid_Size_X']:>7} x {r['Grid_Size_Y']:>3} x {r['Grid_Size_Z']:>4} wg {r['Workgroup_Size_X']:>4}  {r['Kernel_Name'].split('(')[0][-40:]}")
print(f"{len(rows) - i0} kernels, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us")
