"""Does capturing the forward pass in a hipGraph change the step time?  (launch-gap experiment)"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import synthetic_bscans
from defectdetection_viaobjectdetection_amd._capi import check, lib
from defectdetection_viaobjectdetection_amd.engine import SegEngine
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
B = 32
eng = SegEngine("s", 1, (640, 640), max_batch=B)
eng.load_state_dict(synthetic_state_dict("s", 1, seed=0))
imgs = torch.from_numpy(synthetic_bscans(B, seed=1000)).cuda()
preds = torch.empty((B, eng.num_anchors, eng.pred_width), dtype=torch.float32, device="cuda")
protos = torch.empty((B, eng.proto_hw[0], eng.proto_hw[1], 32), dtype=torch.float16, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
def fwd(stream):
    check(lib.m355_forward(eng._h, P(imgs), B, P(preds), P(protos), C.c_void_p(stream.cuda_stream)), eng._h)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): fwd(s)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): fwd(s)
    s.synchronize()
    eager = (time.perf_counter() - t0) / 50
    ref = preds.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fwd(s)
    for _ in range(5): g.replay()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    s.synchronize()
    graph = (time.perf_counter() - t0) / 50
print(f"forward eager {eager * 1e3:.3f} ms  graph {graph * 1e3:.3f} ms  same output {bool(torch.equal(ref, preds))}")
