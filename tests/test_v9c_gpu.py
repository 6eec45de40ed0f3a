"""YOLOv9c-seg on the HIP engine (SURVEY next row N4: the architecture the reference scripts literally name,
/root/reference/BscanBased/yolo_seg_train.py:7, yolo8_seg_predict.py:4-8) against its CPU oracle: same bounds as the
yolov8-seg variants test (raw maps rel-L2, stated tolerances on the 99th percentile, maxima against the fp16-storage floor
is not available for this graph -- no format oracle yet -- so the maxima are only reported), NMS rows bit-exact on equal
predictions, masks >= 99.5 %, and the reference's predict call shape end to end."""
import os

import numpy as np
import pytest
import torch

import yolov8_seg_oracle as orc
import yolov9c_seg_oracle as o9
from helpers import synthetic_bscans

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("nc,shape,batch", [(1, (320, 320), 3), (1, (640, 640), 2), (3, (256, 384), 1)])
def test_v9c_forward_and_postprocess_parity(nc, shape, batch, cuda_device):
    from defectdetection_viaobjectdetection_amd.engine import SegEngine
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict("9c", nc, seed=2, cls_bias=-2.5)
    eng = SegEngine("9c", nc, shape, max_batch=batch)
    eng.load_state_dict(sd)
    oracle = o9.SegmentationModelV9c(nc)
    oracle.load_state_dict(sd)
    oracle.eval()
    imgs = synthetic_bscans(batch, shape[0], shape[1], seed=5)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    with torch.no_grad():
        raw_l, mc, o_protos = oracle.forward_raw(x)
        o_preds, _ = oracle(x)
    preds, protos = eng.forward(torch.from_numpy(imgs).to(cuda_device))
    raw = eng.raw_head(batch).cpu()
    torch.cuda.synchronize()
    A = o_preds.shape[2]
    assert preds.shape == (batch, A, 4 + nc + 32) and torch.isfinite(preds).all()
    o_raw = torch.cat([r.view(batch, 64 + nc, -1) for r in raw_l], 2)
    o_raw = torch.cat((o_raw, mc), 1).permute(0, 2, 1)
    e_box, e_cls, e_mc = rel_l2(raw[..., :64], o_raw[..., :64]), rel_l2(raw[..., 64:64 + nc], o_raw[..., 64:64 + nc]), rel_l2(raw[..., 64 + nc:], o_raw[..., 64 + nc:])
    e_pr = rel_l2(protos.float().cpu().permute(0, 3, 1, 2), o_protos)
    gp, op = preds.cpu(), o_preds.permute(0, 2, 1)
    dbox = (gp[..., :4] - op[..., :4]).abs().flatten()
    dsc = (gp[..., 4:4 + nc] - op[..., 4:4 + nc]).abs().flatten()
    q = lambda t, f: float(t.kthvalue(max(1, int(t.numel() * f)))[0])
    print(f"v9c nc={nc} {shape} b={batch}: raw box {e_box:.2e} cls {e_cls:.2e} coef {e_mc:.2e} proto {e_pr:.2e} | box px median {q(dbox, .5):.4f} "
          f"p99 {q(dbox, .99):.3f} max {float(dbox.max()):.3f} | score p99 {q(dsc, .99):.2e} max {float(dsc.max()):.2e}")
    assert e_box <= 1e-2 and e_mc <= 1e-2 and e_pr <= 1e-2 and e_cls <= 2e-2
    assert q(dbox, .5) <= 0.05 and q(dbox, .99) <= 0.5 and q(dsc, .99) <= 3e-3
    for conf, iou, max_det in ((0.25, 0.7, 300), (0.05, 0.5, 20)):
        dets, counts, masks = eng.postprocess(preds, protos, conf, iou, max_det)
        torch.cuda.synchronize()
        ref = orc.non_max_suppression(preds.cpu().permute(0, 2, 1).numpy(), nc, conf, iou, max_det)
        tot = agree = 0
        for b in range(batch):
            n = int(counts[b])
            assert n == ref[b].shape[0] and np.array_equal(dets[b, :n].cpu().numpy(), ref[b])
            if n:
                d = dets[b, :n].cpu()
                m = orc.process_mask(protos[b].float().cpu().permute(2, 0, 1), d[:, 6:], d[:, :4], shape)
                tot += m.numel()
                agree += int((masks[b, :n].cpu().bool() == m).sum())
        if tot:
            assert agree / tot >= 0.995
    eng.close()


def test_reference_predict_call_with_a_v9c_model(tmp_path, cuda_device):
    """yolo8_seg_predict.py:5-9 with the architecture its checkpoint path names: YOLO(path) -> predict(png, save=True) -> print."""
    from ultralytics import YOLO
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    m = YOLO("yolov9c-seg.yaml")
    m.set_classes(1, {0: "defect"})
    m.load_state_dict(synthetic_state_dict("9c", 1, seed=2, cls_bias=-2.5))
    path = m.save(str(tmp_path / "yolo9c-seg" / "segmentation320" / "weights" / "best.pt"))
    model = YOLO(path)                                                    # :5
    assert model.scale == "9c" and model.nc == 1
    png = os.path.join(GOLDEN, "bscans", "787-225_01_Ch-0_51.png")
    results = model.predict(png, save=True, project=str(tmp_path / "runs"), name="predict", verbose=False)   # :8
    print(results)                                                        # :9
    assert len(results) == 1 and results[0].orig_shape == (320, 320)
    assert results[0].boxes.data.shape[1] == 6 and os.listdir(str(tmp_path / "runs" / "predict"))
