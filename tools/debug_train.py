import sys, os, torch
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'oracle']
from helpers import synthetic_bscans
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
scale, shape, batch = 'n', (128,160), 2
sd = synthetic_state_dict(scale, 1, seed=3)
eng = TrainEngine(scale, 1, shape, batch); eng.load_state_dict(sd)
imgs = synthetic_bscans(batch, shape[0], shape[1], seed=9)
raw, pr = eng.forward(torch.from_numpy(imgs).cuda())
g = torch.Generator().manual_seed(1)
R1 = torch.randn(raw.shape, generator=g).cuda(); R2 = torch.randn(pr.shape, generator=g).cuda()
S = float(sys.argv[1]) if len(sys.argv)>1 else 1.0
# instrument: run backward op by op
import types
orig_ops = eng.ops
eng.backward(R1*S, R2*S)
torch.cuda.synchronize()
for op in reversed(eng.ops):
    if op['kind']=='conv':
        n=op['name']; sv=eng.saved[n]
        dz=sv['dz']; gw=eng.grads[n+'.conv.weight']
        gd=eng.gtensors[op['dst'].t][..., op['dst'].off:op['dst'].off+op['dst'].c]
        print(f"{n:28s} dY max {float(gd.float().abs().max()):10.3e} finite {bool(torch.isfinite(gd).all())} | dz max {float(dz.float().abs().max()):10.3e} fin {bool(torch.isfinite(dz).all())} | dW max {float(gw.abs().max()):10.3e} fin {bool(torch.isfinite(gw).all())}")
