"""Median per-launch-position kernel time over the last N bench steps of a rocprofv3 kernel trace."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]]
steps = [rows[idx[k]:idx[k + 1]] for k in range(len(idx) - 6, len(idx) - 1)]
n = len(steps[0])
out = []
for j in range(n):
    d = statistics.median((int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"])) / 1e3 for s in steps)
    nm = steps[0][j]["Kernel_Name"]
    for key in ("conv_igemm_kernel", "conv3x3_halo_kernel"):
        if key in nm:
            nm = key.replace("_kernel", "") + nm[nm.find("<"):nm.find(">") + 1]
    out.append((j, d, nm[:40]))
print(" ".join(f"{d:.0f}" for _, d, _ in out))
print("total", sum(d for _, d, _ in out))
