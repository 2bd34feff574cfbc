"""Timeline of ONE isolated batched build from a rocprofv3 --kernel-trace csv (the last complete build of the trace): start / duration /
queue per kernel, relative to the build's first kernel: python scripts/one_build_timeline.py DIR"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# builds start with k_set_ptrs / k_cols_fused at level 0 (largest grid): split at k_set_ptrs
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_set_ptrs")]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>2}  grid {r['Grid_Size_X']:>7} x {r['Grid_Size_Y']:>3} x {r['Grid_Size_Z']:>4}  {r['Kernel_Name'].split('(')[0][-30:]}")
