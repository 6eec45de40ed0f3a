"""GPU augmentation kernel (mosaic + affine + HSV + flip) and the host-side label geometry: exact pixel checks for
identity / translation / flip / mosaic placement, and image-vs-label consistency under the random pipeline."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(tmp_path, n=8, size=96):
    from test_train_api_gpu import make_defect_dataset
    from defectdetection_viaobjectdetection_amd.dataset import SegDataset, read_data_yaml
    cfg = read_data_yaml(make_defect_dataset(str(tmp_path / "ds"), n_train=n, n_val=2, size=size, seed=3))
    return SegDataset(cfg["train"], size, nc=1)


def _plan(src, m, mosaic=False, xc=0, yc=0, flip=False, gains=(1.0, 1.0, 1.0)):
    return dict(src=list(src), xc=xc, yc=yc, m=np.asarray(m, np.float64), flip=flip, gains=np.asarray(gains), mosaic=mosaic, inst=[])


def test_identity_translation_flip_and_mosaic_pixels(tmp_path, cuda_device):
    from defectdetection_viaobjectdetection_amd.augment import Augmenter
    ds = _dataset(tmp_path)
    aug = Augmenter(ds, cuda_device)
    H, W = ds.imgsz
    eye = np.eye(3)
    tr = np.eye(3); tr[0, 2], tr[1, 2] = 5, 3
    xc, yc = 70, 60
    mo = np.eye(3); mo[0, 2], mo[1, 2] = -(xc - W // 2), -(yc - H // 2)     # window of the canvas centred on (xc, yc)
    out = aug.render([_plan([2] * 4, eye), _plan([3] * 4, tr), _plan([2] * 4, eye, flip=True),
                      _plan([0, 1, 2, 3], mo, mosaic=True, xc=xc, yc=yc)]).cpu().numpy()
    src = ds.images
    assert np.array_equal(out[0], src[2])
    assert np.array_equal(out[1][3:, 5:], src[3][:-3, :-5]) and (out[1][:3] == 114).all() and (out[1][:, :5] == 114).all()
    assert np.array_equal(out[2], src[2][:, ::-1])
    hw, hh = W // 2, H // 2
    # output (x, y) = canvas (x + xc - W/2, y + yc - H/2); quadrant k shows source k
    assert np.array_equal(out[3][:hh, :hw], src[0][H - hh:, W - hw:])          # top-left: bottom-right corner of image 0
    assert np.array_equal(out[3][:hh, hw:], src[1][H - hh:, :W - hw])
    assert np.array_equal(out[3][hh:, :hw], src[2][:H - hh, W - hw:])
    assert np.array_equal(out[3][hh:, hw:], src[3][:H - hh, :W - hw])


def test_hsv_gains_change_colour_but_not_gray(tmp_path, cuda_device):
    from defectdetection_viaobjectdetection_amd.augment import Augmenter
    ds = _dataset(tmp_path)
    aug = Augmenter(ds, cuda_device)
    out = aug.render([_plan([1] * 4, np.eye(3), gains=(1.0, 1.0, 0.5)), _plan([1] * 4, np.eye(3), gains=(1.0, 0.0, 1.0))]).cpu().numpy()
    src = ds.images[1].astype(np.float32)
    assert np.abs(out[0].astype(np.float32) - 0.5 * src).max() <= 1.0            # V gain scales every channel
    gray = out[1].astype(np.float32)
    assert np.abs(gray - src.max(2, keepdims=True)).max() <= 1.0                  # S = 0 -> every channel = V = max(R,G,B)


def test_random_pipeline_keeps_labels_on_the_defects(tmp_path, cuda_device):
    from defectdetection_viaobjectdetection_amd.augment import Augmenter
    from defectdetection_viaobjectdetection_amd.dataset import rasterize_polygon
    ds = _dataset(tmp_path, n=12, size=160)
    aug = Augmenter(ds, cuda_device, seed=1, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0)
    H, W = ds.imgsz
    inter = union = 0
    n_inst = 0
    for rep in range(3):
        b = aug.batch(list(range(12)), mosaic_on=True)
        img = b["img"].cpu().numpy().astype(np.int32)
        assert img.shape == (12, H, W, 3) and b["masks"].shape == (12, H // 4, W // 4)
        for k, p in enumerate(b["plans"]):
            defect = (img[k, :, :, 0] > 170) & (img[k, :, :, 2] < 110)            # the generator paints (230, 200, 40)
            lab = np.zeros((H, W), bool)
            for _, q in p["inst"]:
                lab |= rasterize_polygon(q, H, W)
                n_inst += 1
            inter += int((defect & lab).sum())
            union += int((defect | lab).sum())
        assert b["bboxes"].shape[0] == b["cls"].shape[0] == b["batch_idx"].shape[0]
        assert (b["bboxes"] >= 0).all() and (b["bboxes"] <= 1).all()
    assert n_inst > 20 and inter / max(union, 1) > 0.85, (n_inst, inter / max(union, 1))


def test_clip_polygon_and_filtering():
    from defectdetection_viaobjectdetection_amd.augment import clip_polygon
    sq = np.array([[-5.0, -5.0], [10.0, -5.0], [10.0, 10.0], [-5.0, 10.0]])
    c = clip_polygon(sq, 8, 8)
    assert len(c) == 4 and c.min() == 0 and c.max() == 8
    assert len(clip_polygon(sq + 100, 8, 8)) == 0
    tri = clip_polygon(np.array([[4.0, -4.0], [12.0, 4.0], [4.0, 12.0]]), 8, 8)
    area = 0.5 * abs(np.dot(tri[:, 0], np.roll(tri[:, 1], -1)) - np.dot(tri[:, 1], np.roll(tri[:, 0], -1)))
    assert area == pytest.approx(8 * 8 - 2 * 8 - 0.0 - (4 * 8 - 0.5 * 4 * 4 * 2) + 0, abs=64)   # stays inside the window
    assert tri.min() >= 0 and tri.max() <= 8
