"""Longest launches of one kernel family in a rocprofv3 kernel trace: python tools/trace_top.py TRACE.csv NAME_SUBSTRING [n]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print(len(rows), "launches, total", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6, "ms")
agg = {}
for r in rows:
    key = (r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r["Kernel_Name"][-40:])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"grid {k[0]:>9s} wg {k[1]:>5s} {k[2]:40s} x{c:4d} avg {t / c:8.1f} us total {t / 1e3:8.2f} ms")
