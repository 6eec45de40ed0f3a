"""Launch time of conv3x3_planes in single-conv mode (m355_conv2d_fwd, tile 33) vs the slab kernel (tile 25) on the 20 x 20 level."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M355_BNECK_REPS"] = "50"
from defectdetection_viaobjectdetection_amd import _capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h = lambda a: a.ctypes.data_as(C.c_void_p)
for (H, W, ci, co, st) in ((80, 80, 64, 64, 1), (40, 40, 128, 128, 1), (40, 40, 128, 64, 1), (40, 40, 64, 64, 1), (80, 80, 128, 128, 1), (80, 80, 128, 224, 1)):
    x = torch.randn(B, H, W, ci, device="cuda").half()
    y = torch.empty(B, H // st, W // st, co, device="cuda", dtype=torch.float16)
    w = (np.random.default_rng(0).standard_normal((co, ci, 3, 3)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = np.zeros(co, np.float32)
    rc = _capi.lib.m355_conv2d_fwd(C.c_void_p(x.data_ptr()), B, H, W, ci, h(w), h(b), co, 3, st, 1, None, C.c_void_p(y.data_ptr()), 0, 33, None)
    print((H, W, ci, co, st), "rc", rc, f"{2 * 9 * ci * co * B * H * W / st / st / 1e9:.1f} GFLOP")
