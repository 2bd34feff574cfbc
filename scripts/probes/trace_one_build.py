import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather")]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else 30]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 26]:
    print(r["Kernel_Name"][:30].ljust(30), round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1),
          r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "vgpr", r["VGPR_Count"], "scr", r["Scratch_Size"])
