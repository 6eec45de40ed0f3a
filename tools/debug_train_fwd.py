import sys, os, torch
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'oracle']
from helpers import synthetic_bscans
import yolov8_seg_oracle as orc
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
scale, shape, batch = 'n', (320,320), 4
sd = synthetic_state_dict(scale, 1, seed=3)
if len(sys.argv) > 1 and sys.argv[1] == "w16":
    sd = {k: (v.half().float() if v.dim() == 4 else v) for k, v in sd.items()}
eng = TrainEngine(scale, 1, shape, batch); eng.load_state_dict(sd)
oracle = orc.SegmentationModel(scale, 1); oracle.load_state_dict(sd); oracle.train()
outs = {}; zs = {}
for n, m in oracle.named_modules():
    if isinstance(m, orc.Conv):
        m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, o.detach()))
        m.conv.register_forward_hook(lambda mod, i, o, n=n: zs.__setitem__(n, o.detach()))
imgs = synthetic_bscans(batch, shape[0], shape[1], seed=9)
x = torch.from_numpy(imgs.transpose(0,3,1,2).copy()).float()/255
with torch.no_grad(): oracle.forward_raw(x)
eng.forward(torch.from_numpy(imgs).cuda()); torch.cuda.synchronize()
rel=lambda a,b: float((a-b).norm()/(b.norm()+1e-20))
for op in eng.ops:
    if op['kind']!='conv': continue
    n=op['name']; d=op['dst']
    a=eng.tensors[d.t][..., d.off:d.off+d.c].float().cpu().permute(0,3,1,2)
    z=eng.saved[n]['z'].float().cpu().permute(0,3,1,2)
    oz=zs[n]; oo=outs[n]
    if op['res'] is not None:
        r=op['res']; oo = oo + eng.tensors[r.t][..., r.off:r.off+r.c].float().cpu().permute(0,3,1,2)
    mean_e=rel(eng.saved[n]['mean'].cpu(), oz.mean((0,2,3)))
    print(f"{n:26s} z {rel(z,oz):.2e}  out {rel(a,oo):.2e}  mean {mean_e:.2e}  zstd-min {float(oz.std((0,2,3)).min()):.3f}")
