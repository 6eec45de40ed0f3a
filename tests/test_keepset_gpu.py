"""End-to-end keep-set parity at the headline configuration (YOLOv8s-seg, 640x640, batch 8): HIP forward + HIP NMS
against oracle forward + oracle NMS, with SURVEY 8d's margin rule (tests/keepset.py) -- every excepted detection
is printed, anything unexplained fails.

Two references: the fp32 oracle and the oracle in the engine's number format (oracle/engine_format_oracle.py, an
independent CPU implementation of fp16 storage + fp32 sums).  Margins: SURVEY 8d's (score within 2e-3 of `conf`, IoU
within 1e-3 of `iou`) or, where larger, what the fp16 storage format itself costs ON THIS BATCH (format oracle vs fp32
oracle, measured inside the test) times 1.25 -- the HIP path may not be further from the fp32 reference than its
prescribed number format is.  (Two correct fp16-storage implementations of a 60-layer network decorrelate at the ulp
level, DESIGN.md section 2, so the format oracle gets the same margins.)
Reference call: /root/reference/BscanBased/yolo8_seg_predict.py:8 (predict = forward + NMS at conf 0.25, iou 0.7).
"""
import numpy as np
import pytest
import torch

import engine_format_oracle as efo
import yolov8_seg_oracle as orc
from helpers import build_oracle, synthetic_bscans
from keepset import anchors_of, common_order_ok, compare_keepsets

pytestmark = pytest.mark.gpu
CONF, IOU, MAX_DET, B = 0.25, 0.7, 300, 8


@pytest.fixture(scope="module")
def runs(cuda_device):
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("s", 1, seed=0)
    imgs = synthetic_bscans(B, seed=11)
    out = {}
    # "gpu": the engine with the raw head maps kept (im2col output convs + decode launch); "gpu_predict": keep_raw=False, the
    # configuration YOLO.predict and bench.py ship (head_tail launches write the prediction rows directly)
    for key, keep in (("gpu", True), ("gpu_predict", False)):
        eng = SegEngine("s", 1, (640, 640), max_batch=B, keep_raw=keep)
        eng.load_state_dict(sd)
        assert any(o["kernel"].startswith("head_tail") for o in eng.op_infos()) == (not keep)
        preds, protos = eng.forward(torch.from_numpy(imgs).to(cuda_device))
        dets, counts, _ = eng.postprocess(preds, protos, CONF, IOU, MAX_DET, masks=False)
        torch.cuda.synchronize()
        out[key] = dict(preds=preds.cpu().numpy(), dets=dets.cpu().numpy(), counts=counts.cpu().numpy())
        eng.close()
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    for name, model in (("fp32", build_oracle("s", 1, sd)), ("fmt", efo.to_engine_format(build_oracle("s", 1, sd)))):
        with torch.no_grad():
            p, _ = model(x)
        p = p.permute(0, 2, 1).contiguous().numpy()                 # (B, A, 37)
        out[name] = dict(preds=p, dets=orc.non_max_suppression(np.ascontiguousarray(p.transpose(0, 2, 1)), 1, CONF, IOU, MAX_DET))
    return out


def _dev(pa, pb):
    """max |score| and max |box| deviation (px) between two prediction tensors (B, A, 37)."""
    return float(np.abs(pa[..., 4] - pb[..., 4]).max()), float(np.abs(pa[..., :4] - pb[..., :4]).max())


def _pair_iou_dev(pa, pb, conf):
    """max |IoU_a(i, j) - IoU_b(i, j)| over pairs of near-candidates (score > conf - 0.05 on side b) that overlap."""
    from keepset import _iou, _xyxy
    worst = 0.0
    for b in range(pa.shape[0]):
        cand = np.nonzero(pb[b, :, 4] > conf - 0.05)[0][:400]
        ba, bb = _xyxy(pa[b]), _xyxy(pb[b])
        for i in range(len(cand)):
            for j in range(i + 1, len(cand)):
                v = _iou(bb[cand[i]], bb[cand[j]])
                if v > 0.4:
                    worst = max(worst, abs(v - _iou(ba[cand[i]], ba[cand[j]])))
    return worst


def _compare(runs, ref, m_conf, m_iou, side="gpu"):
    g, r = runs[side], runs[ref]
    n_exc, n_det, bad_all = 0, 0, []
    for b in range(B):
        n = int(g["counts"][b])
        kg = anchors_of(g["dets"][b, :n], g["preds"][b])
        kr = anchors_of(r["dets"][b], r["preds"][b])
        n_det += len(kr)
        exc, bad = compare_keepsets(kg, g["preds"][b], kr, r["preds"][b], CONF, IOU, m_conf, m_iou)
        for side, a, why in exc:
            who = "HIP" if side == "a" else ref
            print(f"  excepted: image {b} anchor {a} kept by {who} only, rule '{why}': score HIP {g['preds'][b, a, 4]:.5f} "
                  f"{ref} {r['preds'][b, a, 4]:.5f}")
        n_exc += len(exc)
        bad_all += [(b,) + t for t in bad]
        assert common_order_ok(kg, kr, r["preds"][b][:, 4], m_conf), f"image {b}: order of common detections differs"
        common = [a for a in kg if a in set(kr)]
        rows_g = {a: g["dets"][b, i] for i, a in enumerate(kg)}
        rows_r = {a: r["dets"][b][i] for i, a in enumerate(kr)}
        for a in common:     # class index bit-exact, the 32 coefficients are the anchor's own row on each side
            assert rows_g[a][5] == rows_r[a][5]
    print(f"HIP vs {ref}: {n_det} reference detections over {B} images, {n_exc} excepted (m_conf {m_conf:.2e}, m_iou {m_iou:.2e}), "
          f"{len(bad_all)} unexplained")
    return n_exc, n_det, bad_all


def _floor(runs):
    """What fp16 storage costs on this batch: format oracle vs fp32 oracle (score, box px, pair IoU)."""
    if "floor" not in runs:
        fs, fb = _dev(runs["fmt"]["preds"], runs["fp32"]["preds"])
        runs["floor"] = (fs, fb, _pair_iou_dev(runs["fmt"]["preds"], runs["fp32"]["preds"], CONF))
    return runs["floor"]


@pytest.mark.parametrize("side", ["gpu", "gpu_predict"])
def test_prediction_maxima_within_the_format_floor(runs, side):
    fs, fb, fi = _floor(runs)
    gs, gb = _dev(runs[side]["preds"], runs["fp32"]["preds"])
    hs, hb = _dev(runs[side]["preds"], runs["fmt"]["preds"])
    d = np.abs(runs[side]["preds"][..., 4] - runs["fp32"]["preds"][..., 4]).ravel()
    print(f"over {B}x8400 anchors -- format floor: score max {fs:.2e}, box max {fb:.3f} px, pair IoU max {fi:.2e};  HIP vs fp32: score max "
          f"{gs:.2e} (p99.9 {np.quantile(d, .999):.2e}), box max {gb:.3f} px;  HIP vs format oracle: score {hs:.2e}, box {hb:.3f} px")
    assert gs <= 1.5 * fs and gb <= 1.5 * fb and hs <= 1.5 * fs and hb <= 1.5 * fb      # maxima of a heavy-tailed noise: x 1.5
    assert np.quantile(d, .99) <= 2e-3                       # SURVEY 8d's stated score tolerance, 99 % of the anchors


@pytest.mark.parametrize("side", ["gpu", "gpu_predict"])
@pytest.mark.parametrize("ref", ["fmt", "fp32"])
def test_keepset_margin_rule(runs, ref, side):
    """Exact keep-set / order / class except detections inside the margins; margins = SURVEY 8d's (2e-3, 1e-3) or, where
    larger, what the number format costs on this batch (x 1.25).  Every excepted detection is printed."""
    fs, fb, fi = _floor(runs)
    n_exc, n_det, bad = _compare(runs, ref, max(2e-3, 1.25 * fs), max(1e-3, 1.25 * fi), side)
    assert not bad, bad
    assert n_det >= 4 * B and n_exc <= max(4, n_det // 10)
