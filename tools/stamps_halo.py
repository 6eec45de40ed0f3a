import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M355_STAMPS"] = "/tmp/stamps.bin"
from defectdetection_viaobjectdetection_amd import _capi
P=lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B,H,W,cin,cout,k,tile):
    x=torch.randn(B,H,W,cin,device='cuda').half()
    w=torch.randn(cout,cin,k,k)*0.05; b=torch.zeros(cout)
    y=torch.empty(B,H,W,cout,device='cuda',dtype=torch.float16)
    for _ in range(2):
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x),B,H,W,cin,P(w),P(b),cout,k,1,1,P(None),P(y),0,tile,C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    s=np.fromfile("/tmp/stamps.bin",dtype=np.uint64).reshape(-1,4)
    s=s[s[:,0]>0].astype(np.int64)
    t0=s[:,0].min()
    pro=(s[:,1]-s[:,0]); main=(s[:,2]-s[:,1]); epi=(s[:,3]-s[:,2]); tot=(s[:,3]-s[:,0])
    span=(s[:,3].max()-t0)
    print(f"{(B,H,W,cin,cout)} blocks {len(s)} span {span/100:.1f}us(100MHz ticks?) prologue med {np.median(pro)} main med {np.median(main)} epi med {np.median(epi)} total med {np.median(tot)}; sum tot/span/256 = {tot.sum()/span/256:.2f} blocks per CU concurrently")
run(32,160,160,128,128,3,16)
run(32,80,80,64,64,3,16)
run(32,80,80,128,224,3,16)
