import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import synthetic_bscans
from defectdetection_viaobjectdetection_amd.engine import SegEngine
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
eng = SegEngine("s", 1, (640, 640), max_batch=32)
eng.load_state_dict(synthetic_state_dict("s", 1, seed=0))
imgs = torch.from_numpy(synthetic_bscans(32, seed=2024)).cuda()
p, q = eng.forward(imgs); p = p.clone(); q = q.clone()
p4, q4 = eng.forward(imgs[5:8].contiguous())
torch.cuda.synchronize()
dp = (p4 - p[5:8]).abs(); dq = (q4.float() - q[5:8].float()).abs()
print("preds max diff", float(dp.max()), "n diff", int((dp > 0).sum()), "of", dp.numel(), "per image", [int((dp[i] > 0).sum()) for i in range(3)])
print("protos max diff", float(dq.max()), "n diff", int((dq > 0).sum()), "per image", [int((dq[i] > 0).sum()) for i in range(3)])
