import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for i in range(0, len(rows), n):
    g = rows[i:i + n]
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g)
    k = g[0]["Kernel_Name"]
    k = k[k.find("<"):k.find(">") + 1]
    print(k, "grid", g[0]["Grid_Size_X"], "median us", d[len(d) // 2] / 1e3)
