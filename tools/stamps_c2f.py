"""Per-wave section cycles of the fused C2f kernel (csrc/c2f_c32.hip, M355_C2F_STAMPS): one launch at the model.2 shape.
Usage: python tools/stamps_c2f.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["M355_C2F_STAMPS"] = "/tmp/c2f_stamps.bin"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd import _capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 160
g = torch.Generator().manual_seed(0)
x = (torch.randn((B, H, W, 64), generator=g) * 0.8).half().cuda()
y = torch.empty_like(x)
w = [np.random.default_rng(i).standard_normal(n).astype(np.float32) * 0.05 for i, n in enumerate((32 * 288, 32, 32 * 288, 32, 64 * 96, 64))]
hp = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(3):
    _capi.check(_capi.lib.m355_c2f_c32_fwd(C.c_void_p(x.data_ptr()), B, H, W, hp(w[0]), hp(w[1]), hp(w[2]), hp(w[3]), hp(w[4]), hp(w[5]), 1,
                                           C.c_void_p(y.data_ptr()), None))
st = np.fromfile("/tmp/c2f_stamps.bin", dtype=np.uint64).reshape(-1, 8, 8).astype(np.float64)   # (block, wave, section)
tiles = B * 200 / st.shape[0]
print(f"{st.shape[0]} blocks, {tiles:.1f} tiles per block; cycles per tile and wave (median over blocks)")
names_x = ["step+DMA issue", "reads+MFMA", "SiLU+t writes", "wait own DMA", "barrier"]
names_y = ["step+DMA issue", "deferred out epilogue", "cv2 reads+MFMA", "1x1 over y0,y1", "y2 epilogue+4 MFMA", "wait own DMA", "barrier"]
for wv in range(8):
    names = names_x if wv < 4 else names_y
    med = np.median(st[:, wv, :len(names)], axis=0) / tiles
    print(f"wave {wv} ({'X' if wv < 4 else 'Y'}): total {med.sum():7.0f} | " + " | ".join(f"{n} {v:6.0f}" for n, v in zip(names, med)))
