"""Per-op table of the inference engine: kernel, layer, us per launch, TFLOP/s, GB/s (HIP events per op).
Usage: python tools/op_table.py [scale] [batch] [imgsz] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd.engine import SegEngine  # noqa: E402
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
eng = SegEngine(scale, 1, (S, S), max_batch=B, keep_raw=False)   # the predict path
eng.load_state_dict(synthetic_state_dict(scale, 1, seed=0))
x = torch.from_numpy(np.random.default_rng(0).integers(0, 255, (B, S, S, 3), dtype=np.uint8)).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
eng.set_profiling(True)
for _ in range(steps):
    eng.forward(x)
torch.cuda.synchronize()
ms, cnt = eng.collect_op_times()
infos = eng.op_infos()
tot = 0.0
agg = {}
for oi, m, c in zip(infos, ms, cnt):
    if c == 0:
        continue
    us = m / c * 1e3
    tot += us
    tf = oi["flops"] * B / (us * 1e-6) / 1e12
    gb = oi["bytes"] * B / (us * 1e-6) / 1e9
    print(f"{oi['kernel'][:34]:34s} {oi['layer'][:40]:40s} {us:8.1f} us {tf:7.1f} TF/s {gb:7.0f} GB/s")
    a = agg.setdefault(oi["kernel"], [0.0, 0.0])
    a[0] += us
    a[1] += oi["flops"] * B
print(f"total {tot:.1f} us")
for k, (us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k[:40]:40s} {us:8.1f} us  {fl / (us * 1e-6) / 1e12:7.1f} TF/s")
