"""The margin-rule comparator of tests/keepset.py on crafted predictions (CPU; the GPU test uses it end to end)."""
import numpy as np

import yolov8_seg_oracle as orc
from keepset import anchors_of, common_order_ok, compare_keepsets


def _preds(rows):
    """rows: (cx, cy, w, h, score) -> (A, 37) with distinct coefficient fingerprints."""
    p = np.zeros((len(rows), 37), np.float32)
    for i, r in enumerate(rows):
        p[i, :5] = r
        p[i, 5:] = np.arange(32, dtype=np.float32) + 100 * i
    return p


def _keep(p, conf=0.25, iou=0.7):
    det = orc.non_max_suppression(p.T[None], 1, conf, iou, 300)[0]
    return anchors_of(det, p)


def test_identical_predictions_have_nothing_to_explain():
    p = _preds([(100, 100, 50, 50, 0.9), (102, 100, 50, 50, 0.8), (300, 300, 40, 40, 0.5), (500, 500, 10, 10, 0.1)])
    k = _keep(p)
    assert k == [0, 2]
    assert compare_keepsets(k, p, k, p, 0.25, 0.7, 2e-3, 1e-3) == ([], [])
    assert common_order_ok(k, k, p[:, 4], 2e-3)


def test_score_next_to_conf_is_an_exception_not_a_failure():
    a = _preds([(100, 100, 50, 50, 0.9), (300, 300, 40, 40, 0.2510)])
    b = _preds([(100, 100, 50, 50, 0.9), (300, 300, 40, 40, 0.2495)])
    exc, bad = compare_keepsets(_keep(a), a, _keep(b), b, 0.25, 0.7, 2e-3, 1e-3)
    assert bad == [] and exc == [("a", 1, "conf")]
    b[1, 4] = 0.20                                    # 5e-2 away from conf: a real discrepancy
    exc, bad = compare_keepsets(_keep(a), a, _keep(b), b, 0.25, 0.7, 2e-3, 1e-3)
    assert bad == [("a", 1)]


def test_iou_next_to_threshold_and_order_swap_and_cascade():
    # boxes 0 and 1: IoU = 42.5 / 57.5 = 0.7391 on side a, 0.6949 on side b -> rule "iou" with a wide margin only
    a = _preds([(100, 100, 50, 50, 0.9), (107.5, 100, 50, 50, 0.8)])
    b = _preds([(100, 100, 50, 50, 0.9), (109.0, 100, 50, 50, 0.8)])
    ka, kb = _keep(a), _keep(b)
    assert ka == [0] and kb == [0, 1]
    exc, bad = compare_keepsets(ka, a, kb, b, 0.25, 0.7, 2e-3, 5e-2)
    assert bad == [] and exc == [("b", 1, "iou")]
    exc, bad = compare_keepsets(ka, a, kb, b, 0.25, 0.7, 2e-3, 1e-3)
    assert bad == [("b", 1)]
    # order swap: two overlapping boxes whose scores cross
    a = _preds([(100, 100, 50, 50, 0.801), (102, 100, 50, 50, 0.800)])
    b = _preds([(100, 100, 50, 50, 0.800), (102, 100, 50, 50, 0.801)])
    ka, kb = _keep(a), _keep(b)
    assert ka == [0] and kb == [1]
    exc, bad = compare_keepsets(ka, a, kb, b, 0.25, 0.7, 2e-3, 1e-3)
    assert bad == [] and sorted(exc) == [("a", 0, "order"), ("b", 1, "order")]
    # cascade: on side b box 1 drops below conf (rule conf), so box 2 -- suppressed by 1 on side a -- survives there
    a = _preds([(100, 100, 50, 50, 0.2510), (101, 100, 50, 50, 0.2400)])
    b = _preds([(100, 100, 50, 50, 0.2495), (101, 100, 50, 50, 0.2400)])
    assert compare_keepsets(_keep(a), a, _keep(b), b, 0.25, 0.7, 2e-3, 1e-3) == ([("a", 0, "conf")], [])
    a = _preds([(100, 100, 50, 50, 0.2510), (101, 100, 50, 50, 0.2600), (400, 400, 9, 9, 0.9)])
    a[0, 4], a[1, 4] = 0.60, 0.50                     # 0 suppresses 1 on side a
    b = a.copy()
    b[0, 4] = 0.2490                                  # side b: 0 is 0.35 away from its side-a score -> unexplained
    exc, bad = compare_keepsets(_keep(a), a, _keep(b), b, 0.25, 0.7, 2e-3, 1e-3)
    assert ("a", 0) in bad
