import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(t.data_ptr())
B, H, ci, co = 64, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, H, H, ci, device='cuda').half(); dy = torch.randn(B, H, H, co, device='cuda').half()
dw = torch.empty(co, 3, 3, ci, device='cuda')
for _ in range(3):
    _capi.check(_capi.lib.m355_conv2d_wgrad(P(x), P(dy), B, H, H, ci, co, 3, 1, P(dw), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
