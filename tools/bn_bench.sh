#!/bin/bash
R=$GRAFT_REPO_ROOT; rm -rf /tmp/bnb; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/bnb -o t -- python3 $R/tools/bn_bench.py > /tmp/bnb.log 2>&1 || { tail -5 /tmp/bnb.log; exit 1; }
python3 - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob('/tmp/bnb/**/t_kernel_trace.csv', recursive=True)[0])) if 'bn_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
shapes = [(64,160,160,64),(64,80,80,128),(64,40,40,256),(64,160,160,32)]
per = len(rows) // len(shapes)
for i, sh in enumerate(shapes):
    mb = sh[0]*sh[1]*sh[2]*sh[3]*2/1e6
    grp = rows[i*per:(i+1)*per]
    best = {}
    for r in grp:
        n = r['Kernel_Name'].split('bn_')[1].split('_kernel')[0]
        d = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        best[n] = min(best.get(n, 1e9), d)
    mult = {'stats': 1, 'silu_apply': 2, 'silu_bwd_reduce': 2, 'silu_bwd_apply': 3, 'finalize': 0}
    print(sh, " | ".join(f"{n}: {d:.1f} us" + (f" {mult[n]*mb/d*1e3:.0f} GB/s" if mult.get(n) else "") for n, d in best.items()))
PY
