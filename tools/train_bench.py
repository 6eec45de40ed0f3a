"""Step-time breakdown of the training path on synthetic device-resident batches (BASELINE config 3 shape:
YOLOv8s-seg, batch 64).  Usage: python tools/train_bench.py [scale] [batch] [imgsz] [steps]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd._capi import check, lib  # noqa: E402
from defectdetection_viaobjectdetection_amd.loss import SegCriterion  # noqa: E402
from defectdetection_viaobjectdetection_amd.spec import init_state_dict  # noqa: E402
from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda", 0)
eng = TrainEngine(scale, 1, (S, S), B)
eng.load_state_dict(init_state_dict(scale, 1, seed=0))
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.integers(0, 255, (B, S, S, 3), dtype=np.uint8)).to(dev)
n = 2 * B
bidx = torch.arange(B).repeat_interleave(2).float().to(dev)
boxes = torch.tensor(np.stack([rng.uniform(.3, .7, n), rng.uniform(.3, .7, n), rng.uniform(.1, .3, n), rng.uniform(.1, .3, n)], 1), dtype=torch.float32).to(dev)
masks = torch.zeros(B, S // 4, S // 4, device=dev)
masks[:, 40:80, 40:80] = 1
masks[:, 60:70, 60:70] = 2
# labels as a loader hands them over: the small per-instance tensors on the host, the mask maps already on the device
batch = {"batch_idx": bidx.cpu(), "cls": torch.zeros(n), "bboxes": boxes.cpu(), "masks": masks}
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
m1 = torch.zeros(eng.n_train, device=dev); m2 = torch.zeros(eng.n_train, device=dev); ema = eng.flat_params.clone()
tim = {k: 0.0 for k in ("fwd", "loss", "bwd", "opt", "repack")}
criterion = SegCriterion(1, (S, S))


def tick():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(steps + 2):
    if it == 2:
        tim = {k: 0.0 for k in tim}
        t_all = tick()
    # as a training loop runs it: one host synchronisation per step (reading the loss); phases = device time between events
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    prep = criterion.prepare(batch, B, dev)
    ev[0].record()
    raw, protos = eng.forward(imgs)
    ev[1].record()
    items, d_raw, d_protos = criterion(raw, protos, prep, 128.0)
    ev[2].record()
    eng.backward(d_raw, d_protos)
    ev[3].record()
    check(lib.m355_adamw_step(eng.flat_params.data_ptr(), eng.flat_grads.data_ptr(), m1.data_ptr(), m2.data_ptr(), ema.data_ptr(),
                              eng.group.data_ptr(), eng.n_train, 1e-4, 1e-4, 0.9, 0.999, 1e-8, 5e-4, it + 1, 1 / 128.0, 0.999, st()))
    ev[4].record()
    eng.repack()
    ev[5].record()
    loss = float(items.sum() * B)
    ev[5].synchronize()
    for j, k in enumerate(tim):
        tim[k] += ev[j].elapsed_time(ev[j + 1]) * 1e-3
total = tick() - t_all
print(f"yolov8{scale}-seg train b{B} {S}x{S}: {total / steps * 1e3:.1f} ms/step = {B * steps / total:.1f} img/s; loss {loss:.3f}")
print("  " + "  ".join(f"{k} {v / steps * 1e3:.1f} ms" for k, v in tim.items()))
