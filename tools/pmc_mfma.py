"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass over the serial bench (every kernel alone):
  SQ_VALU_MFMA_BUSY_CYCLES  cycles the matrix pipes were busy, summed over every SIMD of the chip
                            (= 16 x the number of 16x16x32 MFMAs, 32 x the number of 32x32x16 MFMAs: MI355X_MICROARCH.md)
  SQ_BUSY_CYCLES            cycles an SQ (one per shader engine, 32 of them) had work, summed: SQ_BUSY_CYCLES / 32 = the
                            kernel's busy cycles.  (GRBM_GUI_ACTIVE / 8 is kept beside it: it also counts the dispatch's
                            ramp and, for the persistent kernels under the counter pass, reads several times higher.)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 256 CUs x 4 SIMDs)
            = the fraction of the kernel's busy time the matrix pipes were busy, averaged over all 1024 SIMDs;
  x the clock the chip holds (1.7-1.9 GHz under this load, tools/stamps_m32.py) / 2.4 GHz = the fraction of the 2.5 PFLOP/s peak.
Usage: python tools/pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import label

per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    lb = label(r["Kernel_Name"])
    if lb is None:
        continue
    per[lb][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[lb].add(r["Dispatch_Id"])
out = {}
for k in sorted(per):
    n = len(disp[k])
    c = {name: v / n for name, v in per[k].items()}
    cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    o = {"launches": n, **{name: round(v, 1) for name, v in c.items()}}
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        o["kernel_busy_cycles"] = round(cyc, 1)
        o["grbm_cycles"] = round(c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, 1)
        o["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
    out[k] = o
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; per-launch "
                   "averages; mfma_util = MFMA busy cycles / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs); bench.py --serial --steps 2 --warmup 1 --batch 32",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k:34s} launches {v['launches']:4d}  mfma_util {v.get('mfma_util', float('nan')):.3f}  LDS bank conflict cycles {v.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
