"""Device-side parity of the training rows that run as torch ops on PyTorch-ROCm and of the validator (A17).

A15: `loss.segmentation_loss` evaluated ON THE DEVICE -- on seeded head outputs and on the TrainEngine's own outputs --
     against the training oracle on CPU copies of the same tensors: loss, its four items, gradients w.r.t. both inputs.
A17: one trained checkpoint validated twice on the same images: through the HIP `Validator` (engine forward + HIP NMS +
     HIP mask assembly + metrics.py) and through the oracle (fp32 forward + oracle NMS + oracle process_mask + the training
     oracle's matcher / 101-point AP).  SURVEY 8d config 3: |delta mAP50| <= 0.002 (the metric's 0.2 points), box and mask.
Reference call: /root/reference/BscanBased/yolo_seg_train.py:12-19 (train() validates every epoch)."""
import os

import numpy as np
import pytest
import torch

import yolov8_seg_oracle as orc
import yolov8_seg_train_oracle as tro
from helpers import build_oracle, synthetic_bscans
from test_loss_host import _case, _oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,B,nc,imgsz,n_inst,empty", [(0, 8, 1, (320, 320), 3, False), (1, 4, 3, (160, 224), 2, True),
                                                           (2, 16, 1, (160, 160), 1, False)])
def test_loss_on_device_matches_oracle(seed, B, nc, imgsz, n_inst, empty, cuda_device):
    from defectdetection_viaobjectdetection_amd import loss as L
    raw, protos, batch, hw = _case(seed, B, nc, imgsz, n_inst, empty)
    lo, io, gro, gpo = _oracle(raw, protos, batch, hw, nc, imgsz)
    dev = cuda_device
    r = raw.to(dev).requires_grad_(True)
    p = protos.to(dev).requires_grad_(True)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    lp, ip = L.segmentation_loss(r, p, dbatch, nc, imgsz)
    lp.backward()
    torch.cuda.synchronize()
    assert lp.device.type == "cuda" and r.grad.device.type == "cuda"
    assert float(lp) == pytest.approx(float(lo), rel=2e-5)
    np.testing.assert_allclose(ip.cpu().numpy(), io.numpy(), rtol=2e-5, atol=1e-6)
    gr, gp = r.grad.cpu(), p.grad.cpu()
    assert float((gr - gro).norm() / gro.norm()) <= 1e-5 and float((gp - gpo).norm() / gpo.norm()) <= 1e-5
    np.testing.assert_allclose(gr.numpy(), gro.numpy(), rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(gp.numpy(), gpo.numpy(), rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("seed,B,nc,imgsz,n_inst", [(0, 8, 1, (320, 320), 3), (3, 4, 3, (160, 224), 6), (5, 16, 1, (640, 640), 2), (7, 2, 1, (160, 160), 0)])
def test_tal_kernels_equal_the_torch_assignment(seed, B, nc, imgsz, n_inst, cuda_device):
    """Task-aligned assignment as two HIP kernels (csrc/loss_kernels.hip: tal_topk / tal_resolve, m355_tal_assign_launch) against
    loss.assign_targets (the torch restatement of upstream's TaskAlignedAssigner) on the same device tensors: the same foreground set,
    assigned instance and target boxes, target scores to fp32 rounding.  Crowded cases (six overlapping instances per image) exercise
    the multiply-claimed anchors."""
    from defectdetection_viaobjectdetection_amd import loss as L
    raw, _, batch, hw = _case(seed, B, nc, imgsz, max(n_inst, 1))
    dev = cuda_device
    A = raw.shape[1]
    k = L._consts(imgsz, imgsz[0] // 4, imgsz[1] // 4, dev, (7.5, 0.5, 1.5))
    r = raw.to(dev)
    boxes, scores = L._decode_all(r, k["anchors"], k["strides_flat"], nc)
    if n_inst == 0:
        batch = {kk: (v[:0] if kk != "masks" else v) for kk, v in batch.items()}
    gt_cls, gt_boxes, gt_valid = L.pad_targets({kk: v.to(dev) for kk, v in batch.items()}, B, imgsz, dev)
    if gt_boxes.shape[1] == 0:
        return                                                     # (no instance: loss_core takes the torch path)
    # (stable_ties: torch.topk leaves ties among equal -- zero -- metrics unspecified; the kernels take candidates first, lower index first)
    want = L.assign_targets(scores, boxes, k["anchors_px"], gt_cls, gt_boxes, gt_valid, stable_ties=True)
    got = L._assign_targets_device(scores, boxes, k["anchors_px"], gt_cls, gt_boxes, gt_valid)
    torch.cuda.synchronize()
    nfg = int(want[2].sum())
    print(f"B={B} A={A} G={gt_boxes.shape[1]}: {nfg} foreground anchors")
    assert nfg > 0
    assert torch.equal(got[2], want[2])                                        # foreground mask
    assert torch.equal(got[3][want[2]], want[3][want[2]])                      # assigned instance (on the foreground)
    assert torch.equal(got[0][want[2]], want[0][want[2]])                      # target boxes
    torch.testing.assert_close(got[1], want[1], rtol=1e-5, atol=1e-9)          # target scores (CIoU^6 as three multiplies vs torch.pow: a few ulp)


def test_loss_on_the_train_engines_outputs(cuda_device):
    """The same comparison on real head maps: TrainEngine.forward -> loss on the device tensors it returned."""
    from defectdetection_viaobjectdetection_amd import loss as L
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    scale, nc, shape, B = "n", 1, (160, 192), 4
    eng = TrainEngine(scale, nc, shape, B)
    eng.load_state_dict(synthetic_state_dict(scale, nc, seed=3))
    raw, protos = eng.forward(torch.from_numpy(synthetic_bscans(B, shape[0], shape[1], seed=9)).to(cuda_device))
    _, _, batch, hw = _case(7, B, nc, shape, 2)
    r = raw.detach().clone().requires_grad_(True)
    p = protos.detach().float().requires_grad_(True)
    lp, ip = L.segmentation_loss(r, p, {k: v.to(cuda_device) for k, v in batch.items()}, nc, shape)
    lp.backward()
    lo, io, gro, gpo = _oracle(raw.detach().cpu(), protos.detach().float().cpu(), batch, hw, nc, shape)
    assert float(lp) == pytest.approx(float(lo), rel=2e-5)
    np.testing.assert_allclose(ip.cpu().numpy(), io.numpy(), rtol=2e-5, atol=1e-6)
    assert float((r.grad.cpu() - gro).norm() / gro.norm()) <= 1e-5 and float((p.grad.cpu() - gpo).norm() / gpo.norm()) <= 1e-5


def _oracle_validation(model, sd, ds, conf=0.001, iou=0.7, max_det=300):
    """fp32 oracle forward + oracle NMS + oracle process_mask + the training oracle's matcher and AP on dataset `ds`."""
    from defectdetection_viaobjectdetection_amd.dataset import rasterize_polygon
    oracle = build_oracle(model.scale, model.nc, sd)
    H, W = ds.imgsz
    tp_b, tp_m, confs, pcls, gcls = [], [], [], [], []
    x = torch.from_numpy(ds.images.transpose(0, 3, 1, 2).copy()).float() / 255.0
    with torch.no_grad():
        preds, protos = oracle(x)
    dets = orc.non_max_suppression(preds.numpy(), model.nc, conf, iou, max_det)
    for i, d in enumerate(dets):
        inst = ds.labels[i]
        g_cls = np.array([c for c, _ in inst], np.int64)
        g_box = np.array([[p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()] for _, p in inst], np.float64).reshape(-1, 4)
        g_msk = np.stack([rasterize_polygon(p, H, W) for _, p in inst]) if inst else np.zeros((0, H, W), bool)
        gcls.append(g_cls)
        if d.shape[0] == 0:
            continue
        m = orc.process_mask(protos[i], torch.from_numpy(d[:, 6:]), torch.from_numpy(d[:, :4]), (H, W)).numpy().astype(bool)
        b = d[:, :4].astype(np.float64)
        ix1 = np.maximum(g_box[:, None, 0], b[None, :, 0]); iy1 = np.maximum(g_box[:, None, 1], b[None, :, 1])
        ix2 = np.minimum(g_box[:, None, 2], b[None, :, 2]); iy2 = np.minimum(g_box[:, None, 3], b[None, :, 3])
        inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
        ag = (g_box[:, 2] - g_box[:, 0]) * (g_box[:, 3] - g_box[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        iou_b = inter / (ag[:, None] + ab[None, :] - inter + 1e-7)
        gm, pm = g_msk.reshape(len(g_msk), H * W).astype(np.float64), m.reshape(len(m), H * W).astype(np.float64)
        im = gm @ pm.T
        iou_m = im / (gm.sum(1)[:, None] + pm.sum(1)[None, :] - im + 1e-7)
        c = d[:, 5].astype(np.int64)
        tp_b.append(tro.match_predictions(c, g_cls, iou_b))
        tp_m.append(tro.match_predictions(c, g_cls, iou_m))
        confs.append(d[:, 4]); pcls.append(c)
    gt_all = np.concatenate(gcls)
    conf_all, cls_all = np.concatenate(confs), np.concatenate(pcls)
    ap_b, _ = tro.ap_per_class(np.concatenate(tp_b), conf_all, cls_all, gt_all)
    ap_m, _ = tro.ap_per_class(np.concatenate(tp_m), conf_all, cls_all, gt_all)
    return {"mAP50(B)": float(ap_b[:, 0].mean()), "mAP50-95(B)": float(ap_b.mean()), "mAP50(M)": float(ap_m[:, 0].mean()),
            "mAP50-95(M)": float(ap_m.mean()), "n_pred": int(conf_all.size)}


def test_validator_map_matches_the_oracle_pipeline(tmp_path, cuda_device):
    from ultralytics import YOLO
    from defectdetection_viaobjectdetection_amd.dataset import SegDataset, read_data_yaml
    from defectdetection_viaobjectdetection_amd.train import Validator
    from test_train_api_gpu import make_defect_dataset
    data = make_defect_dataset(str(tmp_path / "data-seg"), n_train=160, n_val=24, size=160, seed=5)
    # the reference's own B-scans join the validation split (no defects labelled: they contribute false-positive pressure only)
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    from PIL import Image
    k = 0
    for root_, _, files in os.walk(golden):
        for f in sorted(files):
            if f.endswith(".png") and k < 4:
                Image.open(os.path.join(root_, f)).convert("RGB").resize((160, 160)).save(
                    os.path.join(str(tmp_path / "data-seg"), "images", "val", f"fixture_{k}.png"))
                open(os.path.join(str(tmp_path / "data-seg"), "labels", "val", f"fixture_{k}.txt"), "w").close()
                k += 1
    assert k == 4
    model = YOLO("yolov8n-seg.yaml")
    model.train(data=data, epochs=60, imgsz=160, batch=16, project=str(tmp_path / "runs"), name="v", device=0, warmup_epochs=1.0,
                verbose=False)
    best = YOLO(os.path.join(str(tmp_path / "runs"), "v", "weights", "best.pt"))
    cfg = read_data_yaml(data)
    ds = SegDataset(cfg["val"], 160, nc=1)
    assert len(ds) == 28
    v = Validator(ds, best.scale, best.nc, 0, batch=16)
    hip = v(best.state_dict)
    v.close()
    ref = _oracle_validation(best, best.state_dict, ds)
    print("HIP validator:", {k: round(val, 4) for k, val in hip.items() if "mAP" in k})
    print("oracle pipeline:", {k: round(val, 4) if isinstance(val, float) else val for k, val in ref.items()})
    assert ref["mAP50(B)"] > 0.3, "the checkpoint must have learnt something for the comparison to mean anything"
    for key in ("mAP50(B)", "mAP50(M)"):
        assert abs(hip[f"metrics/{key}"] - ref[key]) <= 0.002, (key, hip[f"metrics/{key}"], ref[key])
    for key in ("mAP50-95(B)", "mAP50-95(M)"):
        assert abs(hip[f"metrics/{key}"] - ref[key]) <= 0.005, (key, hip[f"metrics/{key}"], ref[key])


def test_criterion_on_prepared_targets_equals_the_torch_op_loss(cuda_device):
    """SegCriterion (targets padded by ``prepare`` before the forward, HIP kernels for the box / DFL / mask terms, fp16
    prototypes) against ``segmentation_loss`` on fp32 clones of the same inputs: same items, same gradients, across calls with
    different inputs, a changing loss scale and two target widths."""
    from defectdetection_viaobjectdetection_amd import loss as L
    B, imgsz, nc = 4, (128, 160), 1
    A = sum((imgsz[0] // s) * (imgsz[1] // s) for s in (8, 16, 32))
    gl = L.SegCriterion(nc, imgsz)
    g = torch.Generator().manual_seed(3)
    for trial, n_per in enumerate(((2, 1, 2, 2), (1, 2, 2, 0), (3, 1, 0, 2))):          # G = 2, 2, 3
        raw = torch.randn((B, A, 64 + nc + 32), generator=g).to(cuda_device)
        protos = torch.randn((B, imgsz[0] // 4, imgsz[1] // 4, 32), generator=g).half().to(cuda_device)
        bidx = torch.tensor([b for b, n in enumerate(n_per) for _ in range(n)], dtype=torch.float32)
        n = len(bidx)
        cxy = torch.rand((n, 2), generator=g) * 0.5 + 0.25
        wh = torch.rand((n, 2), generator=g) * 0.2 + 0.1
        masks = torch.zeros((B, imgsz[0] // 4, imgsz[1] // 4), dtype=torch.uint8)
        masks[:, 8:16, 10:20] = 1
        masks[:, 12:14, 12:16] = 2
        batch = {"batch_idx": bidx.to(cuda_device), "cls": torch.zeros(n, device=cuda_device), "bboxes": torch.cat((cxy, wh), 1).to(cuda_device),
                 "masks": masks.to(cuda_device)}
        scale = 64.0 * (trial + 1)
        items, d_raw, d_pr = gl(raw, protos, gl.prepare(batch, B, cuda_device), scale)
        r = raw.clone().requires_grad_(True)
        p = protos.float().requires_grad_(True)
        loss, items_ref = L.segmentation_loss(r, p, batch, nc, imgsz)
        (loss * scale).backward()
        torch.cuda.synchronize()
        assert torch.allclose(items, items_ref, rtol=1e-5, atol=1e-6), (items, items_ref)
        assert torch.allclose(d_raw, r.grad, rtol=1e-4, atol=1e-6 * scale)
        assert torch.allclose(d_pr.float(), p.grad, rtol=2e-3, atol=2e-6 * scale)   # fp16 prototype gradient (the engine's dtype)


def test_loss_kernels_equal_the_torch_expressions(cuda_device, monkeypatch):
    """csrc/loss_kernels.hip (all-anchor decode, box / DFL slots, mask term) through the loss against the torch-op forms of the
    same terms on the same device tensors (fp16
    prototypes as the engine hands them over): items, d raw, d protos; and the kernel twice = the same bits."""
    from defectdetection_viaobjectdetection_amd import loss as L
    for seed, B, nc, imgsz, n_inst, empty in ((5, 6, 1, (192, 256), 3, False), (6, 3, 3, (128, 128), 5, True)):
        raw, protos, batch, _ = _case(seed, B, nc, imgsz, n_inst, empty)
        dbatch = {k: v.to(cuda_device) for k, v in batch.items()}
        res = {}
        for mode in ("kernel", "kernel2", "torch"):
            monkeypatch.setenv("M355_NO_LOSS_KERNELS", "1" if mode == "torch" else "0")
            r = raw.to(cuda_device).requires_grad_(True)
            p = protos.half().to(cuda_device).requires_grad_(True)
            lp, ip = L.segmentation_loss(r, p, dbatch, nc, imgsz)
            (lp * 64.0).backward()
            torch.cuda.synchronize()
            res[mode] = (ip.cpu(), r.grad.cpu(), p.grad.float().cpu())
        assert all(torch.equal(a, b) for a, b in zip(res["kernel"], res["kernel2"]))
        ik, grk, gpk = res["kernel"]
        it, grt, gpt = res["torch"]
        np.testing.assert_allclose(ik.numpy(), it.numpy(), rtol=2e-5, atol=1e-7)
        assert float((grk - grt).norm() / grt.norm()) <= 1e-5
        # d protos leaves in fp16 on both paths: one rounding each
        assert float((gpk - gpt).norm() / gpt.norm()) <= 1e-3
        assert float(gpt.abs().max()) > 0


def test_mask_loss_abi_edge_cases(cuda_device):
    """m355_mask_loss_launch directly: boxes partly / wholly outside the map, an empty box, a zero-weight slot with garbage in
    its box, fp32 and fp16 prototypes -- against the dense torch expression in float64."""
    import ctypes as C
    from defectdetection_viaobjectdetection_amd import _capi as capi
    g = torch.Generator().manual_seed(1)
    B, K, mh, mw = 2, 6, 24, 40
    coef = torch.randn(B, K, 32, generator=g)
    masks = torch.randint(0, 4, (B, mh, mw), generator=g, dtype=torch.int32)
    inst = torch.randint(1, 4, (B, K), generator=g, dtype=torch.int32)
    boxes = torch.tensor([[3.2, 2.0, 17.5, 9.9], [-5.0, -3.0, 12.0, 30.0], [35.5, 20.1, 60.0, 24.0], [10.0, 10.0, 10.0, 12.0],
                          [50.0, 5.0, 70.0, 9.0], [float("nan"), 0.0, 1e30, -1e30]]).repeat(B, 1, 1)
    w = torch.rand(B, K, generator=g) + 0.5
    w[:, 5] = 0.0
    cols = torch.arange(mw, dtype=torch.float32).repeat(mh)[None, None]
    rows = torch.arange(mh, dtype=torch.float32).repeat_interleave(mw)[None, None]
    bx = boxes[..., None]
    inside = (cols >= bx[:, :, 0]) & (cols < bx[:, :, 2]) & (rows >= bx[:, :, 1]) & (rows < bx[:, :, 3])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for f16 in (0, 1):
        protos = torch.randn(B, mh, mw, 32, generator=g)
        protos = protos.half() if f16 else protos
        c64 = coef.double().requires_grad_(True)
        p64 = protos.double().requires_grad_(True)
        pred = torch.bmm(c64, p64.reshape(B, mh * mw, 32).transpose(1, 2))
        gt = (masks.reshape(B, 1, -1) == inst[..., None]).double()
        bce = torch.nn.functional.binary_cross_entropy_with_logits(pred, gt, reduction="none")
        ssum = (bce * inside).sum(2)
        (ssum * w.double() * (w != 0)).sum().div(mh * mw).backward()
        d = lambda t: t.contiguous().to(cuda_device)  # noqa: E731
        dc, dp, dm, di, db, dw = d(coef), d(protos), d(masks), d(inst), d(boxes), d(w)
        o_sum = torch.full((B, K), float("nan"), device=cuda_device)
        o_dc = torch.full((B, K, 32), float("nan"), device=cuda_device)
        o_dp = torch.full((B, mh, mw, 32), float("nan"), device=cuda_device)
        capi.check(capi.lib.m355_mask_loss_launch(dc.data_ptr(), dp.data_ptr(), f16, dm.data_ptr(), di.data_ptr(), db.data_ptr(),
                                                  dw.data_ptr(), B, K, mh, mw, o_sum.data_ptr(), o_dc.data_ptr(), o_dp.data_ptr(), 0, None, st))
        # the prototype gradient alone, scaled by a device scalar, stored as fp16
        gsc = torch.tensor(48.0, device=cuda_device)
        o_dp16 = torch.full((B, mh, mw, 32), float("nan"), dtype=torch.float16, device=cuda_device)
        capi.check(capi.lib.m355_mask_loss_launch(dc.data_ptr(), dp.data_ptr(), f16, dm.data_ptr(), di.data_ptr(), db.data_ptr(),
                                                  dw.data_ptr(), B, K, mh, mw, None, None, o_dp16.data_ptr(), 1, gsc.data_ptr(), st))
        torch.cuda.synchronize()
        want_sum = (ssum * (w != 0)).detach()
        np.testing.assert_allclose(o_sum.cpu().double().numpy(), want_sum.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o_dc.cpu().double().numpy(), c64.grad.numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(o_dp.cpu().double().numpy(), p64.grad.numpy(), rtol=1e-4, atol=1e-8)
        assert torch.equal(o_dp16.cpu(), (o_dp * 48.0).half().cpu())
        assert float(o_sum[:, 3].abs().max()) == 0.0 and float(o_sum[:, 4].abs().max()) == 0.0      # empty / outside boxes
    assert capi.lib.m355_mask_loss_launch(dc.data_ptr(), dp.data_ptr(), 1, dm.data_ptr(), di.data_ptr(), db.data_ptr(), dw.data_ptr(),
                                          B, 0, mh, mw, o_sum.data_ptr(), o_dc.data_ptr(), o_dp.data_ptr(), 0, None, st) != 0
    assert capi.lib.m355_mask_loss_launch(dc.data_ptr(), dp.data_ptr(), 1, dm.data_ptr(), di.data_ptr(), db.data_ptr(), dw.data_ptr(),
                                          B, K, mh, mw, None, None, None, 0, None, st) != 0                 # nothing to compute


def test_box_loss_abi_against_autograd(cuda_device):
    """m355_box_loss_launch against autograd through loss.ciou / the DFL expression in float64 on the same slots: overlapping,
    disjoint, containing and degenerate target boxes, target distances beyond the 15-bin range, a zero-weight slot with NaN
    logits (skipped, outputs 0)."""
    import ctypes as C
    from defectdetection_viaobjectdetection_amd import _capi as capi
    from defectdetection_viaobjectdetection_amd import loss as L
    g = torch.Generator().manual_seed(21)
    n = 300
    logits = torch.randn(n, 4, 16, generator=g) * 2
    anc = torch.rand(n, 2, generator=g) * 30 + 5
    ctr = anc + (torch.rand(n, 2, generator=g) - 0.5) * 6
    half = torch.rand(n, 2, generator=g) * 8 + 0.3
    tgt = torch.cat((ctr - half, ctr + half), 1)
    tgt[:20] += 40.0                                                       # disjoint from any prediction, distances clamp to 0 / 14.99
    tgt[20:30, 2:] = tgt[20:30, :2]                                        # zero-area targets
    w = torch.rand(n, generator=g) + 0.1
    w[-3:] = 0.0
    lg64 = logits.double().requires_grad_(True)
    bins = torch.arange(16, dtype=torch.float64)
    ltrb = (lg64.softmax(2) * bins).sum(2)
    pred = torch.cat((anc.double() - ltrb[:, :2], anc.double() + ltrb[:, 2:]), 1)
    box = (1.0 - L.ciou(pred, tgt.double())) * w.double()
    dist = torch.cat((anc - tgt[:, :2], tgt[:, 2:] - anc), 1).clamp(0, 16 - 1 - 0.01)
    lo = dist.long()
    logp = lg64.log_softmax(2)
    ce = -logp.gather(2, lo[..., None]).squeeze(2) * (lo + 1 - dist).double() - logp.gather(2, lo[..., None] + 1).squeeze(2) * (dist - lo).double()
    dfl = ce.mean(1) * w.double()
    (g_box,) = torch.autograd.grad(box.sum(), lg64, retain_graph=True)
    (g_dfl,) = torch.autograd.grad(dfl.sum(), lg64)
    logits_dev = logits.clone()
    logits_dev[-1] = float("nan")
    d = lambda t: t.contiguous().to(cuda_device)  # noqa: E731
    dl, da, dt, dw = d(logits_dev), d(anc), d(tgt), d(w)
    o = [torch.full(sh, float("nan"), device=cuda_device) for sh in ((n,), (n,), (n, 64), (n, 64))]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib.m355_box_loss_launch(dl.data_ptr(), da.data_ptr(), dt.data_ptr(), dw.data_ptr(), n, *[t.data_ptr() for t in o], st))
    torch.cuda.synchronize()
    ob, od, ogb, ogd = [t.cpu().double() for t in o]
    np.testing.assert_allclose(ob.numpy(), box.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(od.numpy(), dfl.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ogb.numpy(), g_box.reshape(n, 64).numpy(), rtol=5e-4, atol=2e-6)
    np.testing.assert_allclose(ogd.numpy(), g_dfl.reshape(n, 64).numpy(), rtol=5e-4, atol=2e-6)
    assert float(ogb[-3:].abs().max()) == 0.0 and float(ob[-3:].abs().max()) == 0.0


@pytest.mark.parametrize("B,A,nc,nm", [(3, 8400, 1, 32), (2, 525, 80, 32), (1, 70, 3, 0)])
def test_dfl_decode_abi_equals_the_torch_expression(B, A, nc, nm, cuda_device):
    """m355_dfl_decode_launch against `(softmax * bins).sum()`, anchor -/+ distance, * stride and sigmoid on the device."""
    import ctypes as C
    from defectdetection_viaobjectdetection_amd import _capi as capi
    g = torch.Generator().manual_seed(2)
    rw = 64 + nc + nm
    raw = (torch.randn(B, A, rw, generator=g) * 3).to(cuda_device)
    anchors = (torch.rand(A, 2, generator=g) * 80).to(cuda_device)
    strides = torch.tensor([8.0, 16.0, 32.0])[torch.randint(0, 3, (A,), generator=g)].to(cuda_device)
    boxes = torch.full((B, A, 4), float("nan"), device=cuda_device)
    scores = torch.full((B, A, nc), float("nan"), device=cuda_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib.m355_dfl_decode_launch(raw.data_ptr(), B * A, A, rw, nc, anchors.data_ptr(), strides.data_ptr(), boxes.data_ptr(),
                                               scores.data_ptr(), st))
    torch.cuda.synchronize()
    bins = torch.arange(16, dtype=torch.float32, device=cuda_device)
    ltrb = (raw[..., :64].view(B, A, 4, 16).softmax(3) * bins).sum(3)
    want = torch.cat((anchors - ltrb[..., :2], anchors + ltrb[..., 2:]), -1) * strides[:, None]
    np.testing.assert_allclose(boxes.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), raw[..., 64:64 + nc].sigmoid().cpu().numpy(), rtol=2e-6, atol=1e-7)
