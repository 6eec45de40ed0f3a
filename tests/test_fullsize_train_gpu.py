"""BASELINE configs 3 and 4 AT FULL SIZE: one training step of YOLOv8s-seg and of YOLOv8m-seg at batch 64,
640 x 640, through the HIP path -- the shapes at which the dispatcher picks the large-map kernels (16 x 16-pixel wide tiles at
160 x 160 / 80 x 80, the patch weight-gradient kernel's large-map split plans) that the 320-pixel parity cases never reach.

  * the step (forward, loss + its backward, backward) runs twice from the same weights: finite loss, no non-finite gradient,
    flat gradient buffer and head outputs BITWISE equal over the two runs (no float atomics anywhere);
  * the loss of the full-size head outputs equals the training oracle's loss on CPU copies of the same tensors;
  * at 640 x 640 and a batch the CPU oracle can afford (2 images, same large maps, same kernel families) the parameter
    gradients are held to the emulated-fp16-storage floor exactly like tests/test_train_engine_gpu.py does at 320.
Reference call: /root/reference/BscanBased/yolo_seg_train.py:12-19 (train(), batch and imgsz per BASELINE configs 3 / 4)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_loss_host import _oracle
from test_train_engine_gpu import _oracle_grads, rel_l2

pytestmark = pytest.mark.gpu


def _batch(B, S, dev, seed=0):
    """Random images + three instances per image sized for the three head levels: a freshly initialised DFL head predicts boxes of
    about 15 strides (the expectation of 16 uniform bins on either side), and the task-aligned metric (CIoU^6) puts the top-10 of a
    ground-truth box on the level whose predictions have its size -- 120, 240 and 480 pixels +- 15 % give every level foreground
    anchors, so EVERY parameter of the head must receive a gradient (round 3 let the box / coefficient branches of up to two levels
    stay at zero because its two random boxes per image could leave a level without foreground)."""
    rng = np.random.default_rng(seed)
    imgs = torch.from_numpy(rng.integers(0, 255, (B, S, S, 3), dtype=np.uint8)).to(dev)
    hw = [(S // s, S // s) for s in (8, 16, 32)]
    mh = mw = S // 4
    bidx, cls, boxes = [], [], []
    masks = torch.zeros(B, mh, mw)
    for b in range(B):
        items = []
        for stride in (8, 16, 32):
            w, h = (15.0 * stride / S * rng.uniform(0.85, 1.15, 2)).tolist()
            w, h = min(w, 0.96), min(h, 0.96)
            cx = float(rng.uniform(w / 2 + 0.01, 1 - w / 2 - 0.01))
            cy = float(rng.uniform(h / 2 + 0.01, 1 - h / 2 - 0.01))
            items.append((w * h, cx, cy, w, h))
        items.sort(reverse=True)                                   # overlap encoding: small instances on top
        for i, (_, cx, cy, w, h) in enumerate(items):
            bidx.append(b); cls.append(0); boxes.append([cx, cy, w, h])
            x1, x2 = int((cx - w / 2) * mw), int((cx + w / 2) * mw)
            y1, y2 = int((cy - h / 2) * mh), int((cy + h / 2) * mh)
            masks[b, y1:y2 + 1, x1:x2 + 1] = i + 1
    batch = {"batch_idx": torch.tensor(bidx, dtype=torch.float32), "cls": torch.tensor(cls, dtype=torch.float32).view(-1, 1),
             "bboxes": torch.tensor(boxes, dtype=torch.float32).view(-1, 4), "masks": masks}
    return imgs, batch, hw


@pytest.mark.parametrize("scale", ["s", "m"])
def test_one_full_size_training_step(scale, cuda_device):
    from defectdetection_viaobjectdetection_amd._capi import check, lib
    from defectdetection_viaobjectdetection_amd.loss import SegCriterion
    from defectdetection_viaobjectdetection_amd.spec import init_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    B, S = 64, 640
    eng = TrainEngine(scale, 1, (S, S), B)
    eng.load_state_dict(init_state_dict(scale, 1, seed=0))
    imgs, batch, hw = _batch(B, S, cuda_device)
    crit = SegCriterion(1, (S, S))
    outs = []
    for _ in range(2):
        prep = crit.prepare(batch, B, cuda_device)
        raw, protos = eng.forward(imgs, update_running_stats=False)
        items, d_raw, d_protos = crit(raw, protos, prep, 128.0)
        eng.backward(d_raw, d_protos)
        ws = torch.zeros(int(lib.m355_grad_sumsq_workspace_floats()), device=cuda_device)
        check(lib.m355_grad_sumsq(C.c_void_p(eng.flat_grads.data_ptr()), eng.n_train, C.c_void_p(ws.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        outs.append((raw.clone(), protos.clone(), eng.flat_grads.clone(), items.clone(), ws[:2].tolist()))
    (raw, protos, grads, items, (sumsq, bad)) = outs[0]
    loss = float(items.sum()) * B
    print(f"yolov8{scale}-seg b{B} @{S}: loss {loss:.4f} items {[round(float(v), 4) for v in items]}; |grad| {sumsq ** 0.5 / 128.0:.4e} over "
          f"{eng.n_train} parameters, {int(bad)} non-finite")
    assert np.isfinite(loss) and loss > 0 and bad == 0 and sumsq > 0
    assert torch.isfinite(raw).all() and torch.isfinite(protos.float()).all()
    # every parameter tensor received a gradient (a layer whose kernel wrote nothing would stay at the zero fill) -- the box and
    # coefficient branches of all three head levels included: the targets put foreground anchors on every level (_batch)
    dead = [k for k, (o, sh) in eng.layout.items() if o < eng.n_train and not bool(grads[o:o + int(np.prod(sh))].any())]
    assert not dead, dead[:8]
    for o in outs[1:]:
        assert torch.equal(o[0], raw) and torch.equal(o[1], protos)
        assert torch.equal(o[2], grads), int((o[2] != grads).sum())
    # the loss of these full-size head outputs against the training oracle on CPU copies of the same tensors
    lo, io, _, _ = _oracle(raw.cpu(), protos.float().cpu(), batch, hw, 1, (S, S))
    assert loss == pytest.approx(float(lo), rel=5e-5)
    np.testing.assert_allclose(items.cpu().numpy(), io.numpy(), rtol=5e-5, atol=1e-6)


@pytest.mark.parametrize("scale", ["s", "m"])
def test_gradients_at_640_against_the_oracle(scale, cuda_device):
    """640 x 640 maps at the batch the CPU oracle affords: the same kernel families as the full-size step (wide 3x3 tiles on the
    160 / 80-pixel maps, patch weight gradients on large maps), gradients held to the fp16-storage floor."""
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from defectdetection_viaobjectdetection_amd.synthetic import synthetic_bscans
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    nc, shape, batch = 1, (640, 640), 2
    sd = synthetic_state_dict(scale, nc, seed=3)
    eng = TrainEngine(scale, nc, shape, batch)
    eng.load_state_dict(sd)
    imgs = synthetic_bscans(batch, shape[0], shape[1], seed=9)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    A = 8400
    g = torch.Generator().manual_seed(1)
    R1 = torch.randn((batch, A, 64 + nc + 32), generator=g)
    R2 = torch.randn((batch, 32, 160, 160), generator=g)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    _, o_raw, protos, g32 = _oracle_grads(scale, nc, sd, x, R1, R2, batch, False)
    _, f_raw, f_protos, g16 = _oracle_grads(scale, nc, sd, x, R1, R2, batch, True)
    raw, pr = eng.forward(torch.from_numpy(imgs).to(cuda_device))
    eng.backward(R1.to(cuda_device), R2.permute(0, 2, 3, 1).contiguous().to(cuda_device))
    torch.cuda.synchronize()
    e_raw, fl_raw = rel_l2(raw.cpu(), o_raw), rel_l2(f_raw, o_raw)
    e_pr, fl_pr = rel_l2(pr.float().cpu().permute(0, 3, 1, 2), protos), rel_l2(f_protos, protos)
    assert e_raw <= 1.5 * fl_raw + 2e-3 and e_pr <= 1.5 * fl_pr + 2e-3
    hip, floor = [], []
    for name, p, gr in eng.trainable():
        got = gr.cpu()
        if got.dim() == 4 and not name.endswith("upsample.weight"):
            got = got.permute(0, 3, 1, 2)
        assert torch.isfinite(got).all(), name
        hip.append(rel_l2(got, g32[name])); floor.append(rel_l2(g16[name], g32[name]))
    hip, floor = np.array(hip), np.array(floor)
    print(f"yolov8{scale}-seg 640x640 b{batch}: forward raw {e_raw:.2e} (floor {fl_raw:.2e}); {len(hip)} gradient tensors: rel-L2 median HIP "
          f"{np.median(hip):.2e} floor {np.median(floor):.2e}, max HIP {hip.max():.2e} floor {floor.max():.2e}")
    assert np.median(hip) <= 1.5 * np.median(floor) + 2e-3
    assert (hip <= 2.5 * np.maximum(floor, np.median(floor)) + 5e-3).all()
