"""How busy is the GPU in a rocprofv3 kernel trace?  union = time with at least one kernel resident, conc = sum of kernel durations / union.
python tools/trace_union.py TRACE.csv [skip_fraction]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4       # window inside the run: steady state
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
lo, hi = t0 + (t1 - t0) * f0, t0 + (t1 - t0) * f1
rows = [r for r in rows if lo <= r[0] < hi]
cur_s, cur_e, union, total = rows[0][0], rows[0][1], 0, 0
gaps = []
for s, e, _ in rows:
    total += e - s
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
wall = rows[-1][1] - rows[0][0]
print(f"wall {wall / 1e6:.2f} ms, union busy {union / 1e6:.2f} ms ({100 * union / wall:.1f} %), sum of kernel durations {total / 1e6:.2f} ms (mean concurrency {total / union:.2f}); "
      f"{len(gaps)} idle gaps, total {sum(gaps) / 1e6:.2f} ms, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, largest {sorted(gaps)[-3:] if gaps else []}")
agg = {}
for s_, e_, n in rows:
    k = n.split("(")[0][-48:]
    agg[k] = agg.get(k, 0) + e_ - s_
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {k:48s} {v / 1e6:8.2f} ms  {100 * v / total:5.1f} %")
