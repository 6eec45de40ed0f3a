"""Experiment: one engine at batch 32 vs two engines at batch 16 on two streams (forward only)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd.engine import SegEngine
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
from defectdetection_viaobjectdetection_amd._capi import check, lib
sd = synthetic_state_dict("s", 1, seed=0)
P = lambda t: C.c_void_p(t.data_ptr())
x = torch.from_numpy(np.random.default_rng(0).integers(0, 255, (32, 640, 640, 3), dtype=np.uint8)).cuda()
def bufs(eng, B):
    return (torch.empty((B, eng.num_anchors, eng.pred_width), dtype=torch.float32, device="cuda"),
            torch.empty((B, eng.proto_hw[0], eng.proto_hw[1], 32), dtype=torch.float16, device="cuda"))
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
e32 = SegEngine("s", 1, (640, 640), max_batch=32); e32.load_state_dict(sd)
p32, q32 = bufs(e32, 32)
s0 = torch.cuda.current_stream()
t = timeit(lambda: check(lib.m355_forward(e32._h, P(x), 32, P(p32), P(q32), C.c_void_p(s0.cuda_stream)), e32._h))
print(f"one engine  b32: {t:.3f} ms / 32 images")
ea = SegEngine("s", 1, (640, 640), max_batch=16); ea.load_state_dict(sd)
eb = SegEngine("s", 1, (640, 640), max_batch=16); eb.load_state_dict(sd)
pa, qa = bufs(ea, 16); pb, qb = bufs(eb, 16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[:16].contiguous(), x[16:].contiguous()
def two():
    check(lib.m355_forward(ea._h, P(xa), 16, P(pa), P(qa), C.c_void_p(s1.cuda_stream)), ea._h)
    check(lib.m355_forward(eb._h, P(xb), 16, P(pb), P(qb), C.c_void_p(s2.cuda_stream)), eb._h)
t = timeit(two)
print(f"two engines b16 on two streams: {t:.3f} ms / 32 images")
def two_seq():
    check(lib.m355_forward(ea._h, P(xa), 16, P(pa), P(qa), C.c_void_p(s1.cuda_stream)), ea._h)
    check(lib.m355_forward(eb._h, P(xb), 16, P(pb), P(qb), C.c_void_p(s1.cuda_stream)), eb._h)
t = timeit(two_seq)
print(f"two engines b16 on one stream: {t:.3f} ms / 32 images")
# two FULL-batch engines alternating on two streams, half a forward out of phase
del ea, eb
eA, eB = e32, SegEngine("s", 1, (640, 640), max_batch=32)
eB.load_state_dict(sd)
pB, qB = bufs(eB, 32)
def fa(): check(lib.m355_forward(eA._h, P(x), 32, P(p32), P(q32), C.c_void_p(s1.cuda_stream)), eA._h)
def fb(): check(lib.m355_forward(eB._h, P(x), 32, P(pB), P(qB), C.c_void_p(s2.cuda_stream)), eB._h)
for off_ms in (0.0, 0.6, 1.1, 1.6):
    torch.cuda.synchronize()
    fa(); torch.cuda.synchronize(); fb(); torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    fa()
    time.sleep(off_ms * 1e-3)
    fb()
    for _ in range(n - 1):
        fa(); fb()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / (2 * n) * 1e3
    print(f"two b32 engines, two streams, B enqueued {off_ms} ms after A: {t:.3f} ms / 32 images")
