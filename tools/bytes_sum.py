import sys; sys.path.insert(0,'/root/repo')
import torch
from defectdetection_viaobjectdetection_amd.engine import SegEngine
eng = SegEngine("s", 1, (640, 640), max_batch=32)
infos = eng.op_infos()
B=32
tb=sum(i["bytes"]*B + i["weight_bytes"] for i in infos); tf=sum(i["flops"]*B for i in infos)
print(f"ops {len(infos)}  algorithmic bytes per forward {tb/1e9:.2f} GB  flops {tf/1e12:.3f} TFLOP")
print(f"HBM floor @5.0 TB/s {tb/5e12*1e3:.2f} ms ; MFMA floor @2.0 PF {tf/2.0e15*1e3:.2f} ms")
by={}
for i in infos:
    k=i["kernel"]; by.setdefault(k,[0,0]); by[k][0]+=i["bytes"]*B+i["weight_bytes"]; by[k][1]+=i["flops"]*B
for k,(b,f) in sorted(by.items(), key=lambda kv:-kv[1][0]): print(f"  {k:34s} {b/1e6:8.0f} MB  {f/1e9:8.0f} GFLOP  floor {max(b/5e12, f/2e15)*1e6:6.0f} us")
