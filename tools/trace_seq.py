"""Kernel sequence of the LAST step in a rocprofv3 kernel trace of tools/train_bench.py: every launch in start order with its
duration and the idle gap before it (python tools/trace_seq.py TRACE.csv [marker_substring]).  The step boundary is the last
launch whose name contains the marker (default: the optimizer kernel); short names."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "adamw_step"
ends = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|m355::|at::native::|void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:90]


prev = int(rows[lo - 1]["End_Timestamp"]) if lo else int(rows[0]["Start_Timestamp"])
busy = idle = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(s - prev, 0)
    busy += e - s
    idle += gap
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap / 1e3:7.1f}  {short(r['Kernel_Name'])}")
    prev = max(prev, e)
print(f"launches {hi - lo}  busy {busy / 1e6:.2f} ms  idle {idle / 1e6:.2f} ms")
