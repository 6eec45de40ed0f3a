#!/bin/bash
# SQ counters of the fused Bottleneck kernel alone (tools/bneck_bench.py): LDS conflicts, MFMA busy, waits.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bneck; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/m -o m -- python3 $R/tools/bneck_bench.py 32 > $O/m.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/m/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60] + " grid " + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    if "planes" not in k: continue
    m = max(n[k], 1)
    print(k, "dispatches", m)
    for name, v in sorted(c.items()): print(f"   {name:28s} {v / m:14.0f}")
    if c.get("SQ_LDS_IDX_ACTIVE"): print(f"   conflict / idx_active = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}; mfma busy / (busy_cycles/32*... ) raw ratio mfma/busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c['SQ_BUSY_CYCLES'],1):.3f}")
PY
