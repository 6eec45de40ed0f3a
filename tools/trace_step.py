"""Print the kernel timeline of the last bench step from a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
for r in rows[a:b]:
    n = r["Kernel_Name"]
    for key in ("conv_igemm_kernel", "conv3x3_halo_kernel"):
        if key in n:
            n = key.replace("_kernel", "") + n[n.find("<"):n.find(">") + 1]
    if "_ZN" in n:
        n = n[n.find("N_1") + 5:][:24]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}us grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} lds={r['LDS_Block_Size']:>6s} vgpr={r['VGPR_Count']:>3s} {n[:60]}")
print("span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "busy us", busy)
