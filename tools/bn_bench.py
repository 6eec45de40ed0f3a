"""Batch-norm training kernels on three activation shapes of YOLOv8s-seg at batch 64 (run under rocprofv3 --kernel-trace, read with
tools/trace_top.py ... bn_): bytes per launch are printed so GB/s follow from the durations."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, Cc) in [(64, 160, 160, 64), (64, 80, 80, 128), (64, 40, 40, 256), (64, 160, 160, 32)]:
    z = torch.randn(B, H, W, Cc, device="cuda").half(); dy = torch.randn_like(z); y = torch.empty_like(z); dz = torch.empty_like(z)
    g = torch.ones(Cc, device="cuda"); b = torch.zeros(Cc, device="cuda"); mean = torch.empty(Cc, device="cuda"); inv = torch.empty(Cc, device="cuda")
    ws = torch.zeros(int(_capi.lib.m355_bn_workspace_floats(Cc)), device="cuda"); dbg = torch.empty(2 * Cc, device="cuda")
    for _ in range(3):
        _capi.check(_capi.lib.m355_bn_silu_train_fwd(P(z), B, H, W, Cc, P(g), P(b), 1e-3, 1, P(y), P(mean), P(inv), P(ws), st))
        _capi.check(_capi.lib.m355_bn_silu_train_bwd(P(z), P(dy), B, H, W, Cc, P(mean), P(inv), P(g), P(b), 1, P(dz), P(dbg), P(ws), st))
    torch.cuda.synchronize()
    print(f"{(B, H, W, Cc)}: tensor {z.numel() * 2 / 1e6:.0f} MB")
