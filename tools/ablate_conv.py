"""Ablation experiment: time the conv kernel with parts switched off (outputs are wrong when dbg != 128).
dbg bits: 1 skip activation LDS-DMA (t>1), 2 skip weight LDS-DMA (t>1), 8 skip barrier, 16 skip ds_reads,
32 skip SiLU, 128 no-op marker (forces the 5-repeat timing path)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd import _capi
P=lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B,H,W,cin,cout,k,tile,dbg):
    x=torch.randn(B,H,W,cin,device='cuda').half()
    w=torch.randn(cout,cin,k,k)*0.05; b=torch.zeros(cout)
    y=torch.empty(B,H,W,cout,device='cuda',dtype=torch.float16)
    _capi.check(_capi.lib.m355_conv2d_fwd(P(x),B,H,W,cin,P(w),P(b),cout,k,1,1,P(None),P(y),0,tile|(dbg<<8),C.c_void_p(torch.cuda.current_stream().cuda_stream)))
for dbg in (128, 128|64, 128|3, 128|3|64):
    run(32,160,160,128,128,3,0,dbg)
    run(32,80,80,64,64,3,1,dbg)
    run(32,80,80,256,128,1,0,dbg)
    run(32,20,20,256,256,3,0,dbg)
