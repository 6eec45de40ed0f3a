"""Per-section cycles of the model.1 + model.2.cv1 patch kernel (conv3x3_s2c32.hip), from s_memtime stamps of every wave.
Usage: M355_S2C32_STAMPS=/tmp/s2.bin python tools/stamps_s2c32.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("M355_S2C32_STAMPS", "/tmp/s2.bin")
import torch
from helpers import synthetic_bscans
from defectdetection_viaobjectdetection_amd.engine import SegEngine
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
eng = SegEngine("s", 1, (640, 640), max_batch=32, keep_raw=False)
eng.load_state_dict(synthetic_state_dict("s", 1, seed=0))
x = torch.from_numpy(synthetic_bscans(32, seed=1)).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
raw = np.fromfile(os.environ["M355_S2C32_STAMPS"], dtype=np.uint64).reshape(-1, 8)
s = raw[:, :6].astype(np.int64)
names = ["decode+offsets+DMA issue", "3x3 reads+MFMA", "3x3 epilogue (SiLU, LDS write)", "1x1 reads+MFMA", "wait next patch + barrier", "1x1 epilogue + stores"]
tot = s.sum(1)
print(f"{len(s)} waves, 25 tiles each: total cycles per wave med {np.median(tot):.0f} = {np.median(tot) / 25:.0f} per tile")
for k, n in enumerate(names):
    print(f"   {n:34s} {np.median(s[:, k]) / 25:8.0f} cycles per tile ({100 * np.median(s[:, k]) / np.median(tot):4.1f} %)")
rt0 = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64); rt1 = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
span = (rt1.max() - rt0.min()) / 100.0
print(f"kernel span by s_memrealtime: {span:.1f} us; wave entry skew: med {np.median(rt0 - rt0.min()) / 100:.1f} us, max {(rt0.max() - rt0.min()) / 100:.1f} us; "
      f"wave lifetime med {np.median(rt1 - rt0) / 100:.1f} us; prologue med {np.median(raw[:, 6].astype(np.int64)):.0f} cycles; "
      f"clock (loop cycles / lifetime): {np.median((tot + raw[:, 6].astype(np.int64)) / np.maximum(rt1 - rt0, 1)) * 100 / 1e3:.2f} GHz")

def span_now():
    r = np.fromfile(os.environ["M355_S2C32_STAMPS"], dtype=np.uint64).reshape(-1, 8)
    a0 = (r[:, 7] & np.uint64(0xffffffff)).astype(np.int64); a1 = (r[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
    cyc = r[:, :7].astype(np.int64).sum(1)
    return (a1.max() - a0.min()) / 100.0, float(np.median(cyc / np.maximum(a1 - a0, 1)) / 10), r[:, :6].astype(np.int64)
spans = []
for i in range(40):
    eng.forward(x)
    torch.cuda.synchronize()
    sp, clk, sec = span_now()
    spans.append((round(sp, 1), round(clk, 2)))
print("spans (us, GHz) of 40 consecutive forwards:", spans)
print("last: per-tile sections", [int(np.median(sec[:, k]) / 25) for k in range(6)])
