"""Ablation timing of the halo conv kernel (outputs wrong when bits 1/2 set): run under rocprofv3 --kernel-trace --stats.
dbg bits: 1 skip patch LDS-DMA in the loop, 2 skip weight LDS-DMA in the loop, 128 marker (5 repeats)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B, H, W, cin, cout, tile, dbg):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, 3, 3) * 0.05; b = torch.zeros(cout)
    y = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, 3, 1, 1, P(None), P(y), 0, tile | (dbg << 8),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [18]
for tile in variants:
    for dbg in (128, 128 | 1, 128 | 2, 128 | 3):
        run(32, 160, 160, 128, 128, tile, dbg)
