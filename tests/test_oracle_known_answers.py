"""Pins the CPU oracle (oracle/yolov8_seg_oracle.py) with closed-form known answers and invariants.

The reference holds no test or golden vector for its YOLO path, so the oracle is
pinned by: exact published parameter counts, conv-MAC totals, output shapes, and per-stage known answers
(letterbox taps, DFL on one-hot logits, dist2bbox, NMS on crafted boxes incl. ties and the IoU boundary,
crop/threshold masks, scale_boxes) plus the reference's own committed B-scan PNGs as inputs.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

import yolov8_seg_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("scale,nc,expected", [
    ("n", 80, 3409968), ("s", 80, 11821056), ("m", 80, 27285968),   # published 3.4 / 11.8 / 27.3 M
    ("n", 1, 3263811), ("s", 1, 11790483), ("m", 1, 27240227),       # SURVEY 8a, nc=1 (data-seg.yaml:4-5)
])
def test_parameter_counts(scale, nc, expected):
    assert orc.count_parameters(orc.SegmentationModel(scale, nc)) == expected


@pytest.mark.parametrize("scale,macs", [("n", 5670220800), ("s", 19957606400)])
def test_conv_macs(scale, macs):
    assert orc.conv_macs_per_image(scale, 1, 640) == macs


def test_output_shapes():
    m = orc.SegmentationModel("n", 1).eval()
    with torch.no_grad():
        preds, protos = m(torch.zeros(2, 3, 640, 640))
        raw, mc, p = m.forward_raw(torch.zeros(1, 3, 320, 320))
    assert preds.shape == (2, 37, 8400) and protos.shape == (2, 32, 160, 160)
    assert [tuple(r.shape) for r in raw] == [(1, 65, 40, 40), (1, 65, 20, 20), (1, 65, 10, 10)]
    assert mc.shape == (1, 32, 2100) and p.shape == (1, 32, 80, 80)


def test_head_bias_init():
    m = orc.SegmentationModel("s", 1)
    for l, s in enumerate((8, 16, 32)):
        assert torch.all(m.model[22].cv2[l][-1].bias == 1.0)
        assert math.isclose(float(m.model[22].cv3[l][-1].bias[0]), math.log(5 / 1 / (640 / s) ** 2), rel_tol=1e-6)
    assert torch.equal(m.model[22].dfl.conv.weight.flatten(), torch.arange(16.0))


def test_letterbox_exact_2x_taps():
    """320 -> 640: r = 2, no padding; interior taps 0.75/0.25, edges replicate (A.2)."""
    row = np.array([0, 100, 200, 40], np.uint8)
    img = np.repeat(np.repeat(row[None, :, None], 4, 0), 3, 2)
    out, r, pad = orc.letterbox(img, (8, 8))
    assert r == 2.0 and pad == (0, 0) and out.shape == (8, 8, 3)
    expect = [0, 25, 75, 125, 175, 160, 80, 40]  # 0.75*a + 0.25*b, rounded
    assert out[3, :, 0].tolist() == expect
    assert np.all(out[:, :, 0] == out[0, :, 0])


def test_letterbox_padding_and_fixture():
    img = np.zeros((100, 200, 3), np.uint8)
    out, r, (left, top) = orc.letterbox(img, (640, 640))
    assert out.shape == (640, 640, 3) and r == 3.2 and (left, top) == (0, 160)
    assert np.all(out[:160] == 114) and np.all(out[480:] == 114) and np.all(out[160:480] == 0)
    from PIL import Image
    png = np.asarray(Image.open(os.path.join(GOLDEN, "bscans", "787-225_01_Ch-0_51.png")).convert("RGB"))
    assert png.shape == (320, 320, 3)           # the file yolo8_seg_predict.py:8 predicts on, mode L, 320x320
    lb, r, pad = orc.letterbox(png, (640, 640))
    assert lb.shape == (640, 640, 3) and r == 2.0 and pad == (0, 0)
    assert np.array_equal(lb[1::2, 1::2][:-1, :-1] // 1, lb[1::2, 1::2][:-1, :-1])  # dtype stays uint8


def test_annotation_fixture_semantics():
    """annotations.json bbox = [x_min, x_max, y_min, y_max] with x often reversed."""
    ann = json.load(open(os.path.join(GOLDEN, "annotations_excerpt.json")))["annotations"]
    b = ann["787-225_01_Ch-0"]["51.png"]
    assert b[0]["bbox"] == [314, 262, 112, 138] and b[1]["bbox"] == [111, 0, 112, 141]
    assert ann["787-226_03_Ch-0"]["81.png"][0]["bbox"] == [233, 81, 53, 77]
    assert all(e["label"] == "Delamination" for seq in ann.values() for fr in seq.values() for e in fr)


def test_dfl_one_hot_and_uniform():
    dfl = orc.DFL(16)
    x = torch.full((1, 64, 3), -1e4)
    for side, k in enumerate((0, 5, 15, 9)):
        x[0, side * 16 + k, :] = 1e4
    out = dfl(x)
    assert torch.allclose(out[0, :, 0], torch.tensor([0.0, 5.0, 15.0, 9.0]))
    assert torch.allclose(dfl(torch.zeros(1, 64, 2)), torch.full((1, 4, 2), 7.5))


def test_anchors_and_decode_known_answer():
    pts, st = orc.make_anchors([(2, 3), (1, 1)], [8, 16])
    assert pts.tolist() == [[0.5, 0.5], [1.5, 0.5], [2.5, 0.5], [0.5, 1.5], [1.5, 1.5], [2.5, 1.5], [0.5, 0.5]]
    assert st.flatten().tolist() == [8.0] * 6 + [16.0]
    # decode: anchor (1.5,0.5), l,t,r,b = 1,0.5,2,1.5 -> x1y1 (0.5,0) x2y2 (3.5,2) -> cxcywh (2,1,3,2) * 8
    a = torch.tensor([1.5, 0.5])
    lt, rb = torch.tensor([1.0, 0.5]), torch.tensor([2.0, 1.5])
    c = ((a - lt) + (a + rb)) / 2
    wh = (a + rb) - (a - lt)
    assert (torch.cat((c, wh)) * 8).tolist() == [16.0, 8.0, 24.0, 16.0]


def _pred(boxes_xywh, scores, nm=2):
    a = len(scores)
    p = np.zeros((1, 4 + 1 + nm, a), np.float32)
    p[0, :4] = np.asarray(boxes_xywh, np.float32).T
    p[0, 4] = scores
    p[0, 5:] = np.arange(a)[None, :]
    return p


def test_nms_crafted_boxes():
    # b0 and b1 overlap with IoU 0.8 (> 0.7: suppressed), b2 far away, b3 below conf
    boxes = [[50, 50, 100, 100], [55, 50, 110, 100], [300, 300, 40, 40], [50, 50, 100, 100]]
    out = orc.non_max_suppression(_pred(boxes, [0.9, 0.8, 0.7, 0.2]), 1, 0.25, 0.7, 300)[0]
    assert out.shape == (2, 8)
    assert out[:, 4].tolist() == pytest.approx([0.9, 0.7])
    assert out[0, :4].tolist() == [0.0, 0.0, 100.0, 100.0] and out[1, :4].tolist() == [280.0, 280.0, 320.0, 320.0]
    assert out[:, 6].tolist() == [0.0, 2.0]  # mask coefficients travel with the kept anchor


def test_nms_threshold_is_strict_and_ties_keep_lower_index():
    # identical scores: lower anchor index first.  IoU exactly 0.5 with thr 0.5 must NOT suppress (IoU > thr).
    boxes = [[50, 50, 100, 100], [100, 50, 100, 100] , [50, 50, 100, 100]]
    iou = orc.box_iou_f32(np.array([0, 0, 100, 100], np.float32), np.array([[50, 0, 150, 100]], np.float32))
    assert float(iou[0]) == pytest.approx(1 / 3)
    out = orc.non_max_suppression(_pred(boxes, [0.6, 0.6, 0.6]), 1, 0.25, 1 / 3 + 1e-6, 300)[0]
    assert out[:, 6].tolist() == [0.0, 1.0]          # anchor 2 duplicates anchor 0 -> suppressed; 1 survives
    out = orc.non_max_suppression(_pred(boxes, [0.6, 0.6, 0.6]), 1, 0.25, 0.2, 300)[0]
    assert out[:, 6].tolist() == [0.0]
    # max_det cap and empty input
    out = orc.non_max_suppression(_pred([[10 + 200 * i, 10, 10, 10] for i in range(5)], [0.9, 0.8, 0.7, 0.6, 0.5]), 1,
                                  0.25, 0.7, 3)[0]
    assert out.shape[0] == 3 and out[:, 4].tolist() == pytest.approx([0.9, 0.8, 0.7])
    assert orc.non_max_suppression(_pred([[1, 1, 1, 1]], [0.1]), 1, 0.25, 0.7, 300)[0].shape == (0, 8)


def test_nms_class_offset():
    """Non-agnostic NMS: identical boxes of different classes both survive (offset cls * 7680)."""
    p = np.zeros((1, 4 + 2 + 1, 2), np.float32)
    p[0, :4, :] = np.array([[50, 50, 100, 100]], np.float32).T
    p[0, 4, 0] = 0.9
    p[0, 5, 1] = 0.8
    out = orc.non_max_suppression(p, 2, 0.25, 0.7, 300)[0]
    assert out[:, 5].tolist() == [0.0, 1.0]


def test_process_mask_crop_and_threshold():
    protos = torch.zeros(2, 8, 8)
    protos[0] = 1.0
    protos[1, :, 4:] = -3.0
    coefs = torch.tensor([[1.0, 0.0], [1.0, 1.0]])
    boxes = torch.tensor([[8.0, 8.0, 24.0, 24.0], [0.0, 0.0, 32.0, 32.0]])  # in a 32x32 input: proto scale 1/4
    m = orc.process_mask(protos, coefs, boxes, (32, 32))
    assert m.shape == (2, 32, 32) and m.dtype == torch.bool
    # det 0: positive logits inside proto cells [2,6) x [2,6) -> bilinear edge at half weight stays > 0
    assert m[0, 12:20, 12:20].all() and not m[0, :6].any() and not m[0, :, 26:].any()
    # det 1: left half +1, right half -2 -> sign flip between proto columns 3 and 4
    assert m[1, :, :14].all() and not m[1, :, 18:].any()


def test_scale_boxes_roundtrip():
    b = np.array([[100.0, 180.0, 300.0, 400.0, 0.9, 0.0]], np.float32)
    out = orc.scale_boxes((640, 640), b[:, :4], (100, 200))  # gain 3.2, pad y 160
    assert out[0].tolist() == pytest.approx([31.25, 6.25, 93.75, 75.0])
    out = orc.scale_boxes((640, 640), np.array([[-5.0, 0.0, 700.0, 700.0]], np.float32), (320, 320))
    assert out[0].tolist() == [0.0, 0.0, 320.0, 320.0]


def test_nms_multi_label_keeps_one_row_per_anchor_class_pair():
    """A11 in the validator's mode: an anchor with two classes above the threshold yields two candidates (same box, class
    offsets keep them from suppressing each other); the best-class mode yields one."""
    import numpy as np
    import yolov8_seg_oracle as orc
    nc, nm = 2, 4
    pred = np.zeros((1, 4 + nc + nm, 3), np.float32)
    pred[0, :4, 0] = (50, 50, 20, 20); pred[0, 4:6, 0] = (0.9, 0.6)        # both classes confident
    pred[0, :4, 1] = (51, 50, 20, 20); pred[0, 4:6, 1] = (0.8, 0.1)        # overlaps anchor 0: suppressed in class 0
    pred[0, :4, 2] = (150, 150, 20, 20); pred[0, 4:6, 2] = (0.2, 0.7)
    one = orc.non_max_suppression(pred, nc, 0.25, 0.5, 300)[0]
    multi = orc.non_max_suppression(pred, nc, 0.25, 0.5, 300, multi_label=True)[0]
    assert one[:, 4].tolist() == [np.float32(0.9), np.float32(0.7)] and one[:, 5].tolist() == [0.0, 1.0]
    assert multi[:, 4].tolist() == [np.float32(0.9), np.float32(0.7), np.float32(0.6)] and multi[:, 5].tolist() == [0.0, 1.0, 1.0]
    assert np.array_equal(multi[0, :4], multi[2, :4])                      # the same anchor twice, once per class
