"""Per-wave section cycles of the model.3 + model.4.cv1 kernel (csrc/conv3x3_s2c64.hip, M355_S2C64_STAMPS) inside one s-seg
forward at batch 32.  Usage: python tools/stamps_s2c64.py"""
import os
import sys

import numpy as np
import torch

os.environ["M355_S2C64_STAMPS"] = "/tmp/s2c64_stamps.bin"
os.environ["M355_NO_LANES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd.engine import SegEngine  # noqa: E402
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict  # noqa: E402

B = 32
eng = SegEngine("s", 1, (640, 640), max_batch=B)
eng.load_state_dict(synthetic_state_dict("s", 1, seed=0))
x = torch.from_numpy(np.random.default_rng(0).integers(0, 255, (B, 640, 640, 3), dtype=np.uint8)).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
st = np.fromfile("/tmp/s2c64_stamps.bin", dtype=np.uint64).reshape(-1, 8, 8).astype(np.float64)
names = ["decode+DMA issue", "S2 (prev tile)", "K", "Z", "barrier 1", "wait patch", "barrier 2"]
print(f"{st.shape[0]} blocks, tiles per block {st[:, :, 7].min():.0f}-{st[:, :, 7].max():.0f}; cycles per tile and wave (median over blocks)")
for wv in range(8):
    med = np.median(st[:, wv, :7] / np.maximum(st[:, wv, 7:8], 1), axis=0)
    print(f"wave {wv}: total {med.sum():7.0f} | " + " | ".join(f"{n} {v:6.0f}" for n, v in zip(names, med)))
