"""Detection / segmentation validation metrics: IoU matching at the ten COCO
thresholds and 101-point interpolated AP.  Stands where upstream's ``SegmentationValidator`` + ``ap_per_class`` stand
(runs inside ``train``; call site /root/reference/BscanBased/yolo_seg_train.py:12).  numpy, a few thousand rows."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

IOU_THRESHOLDS = np.linspace(0.5, 0.95, 10)


def box_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n,4) x (m,4) xyxy -> (n,m)."""
    if a.size == 0 or b.size == 0:
        return np.zeros((a.shape[0], b.shape[0]), np.float64)
    a = a.astype(np.float64)[:, None, :]
    b = b.astype(np.float64)[None, :, :]
    iw = np.clip(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]), 0, None)
    ih = np.clip(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]), 0, None)
    inter = iw * ih
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    return inter / (ua + 1e-7)


def mask_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n,H,W) x (m,H,W) binary -> (n,m): intersection as one matrix product over the flattened pixels."""
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((a.shape[0], b.shape[0]), np.float64)
    fa = a.reshape(a.shape[0], -1).astype(np.float32)
    fb = b.reshape(b.shape[0], -1).astype(np.float32)
    inter = (fa @ fb.T).astype(np.float64)
    union = fa.sum(1, dtype=np.float64)[:, None] + fb.sum(1, dtype=np.float64)[None, :] - inter
    return inter / (union + 1e-7)


def match(pred_cls: np.ndarray, gt_cls: np.ndarray, iou_pg: np.ndarray) -> np.ndarray:
    """iou_pg (n_pred, n_gt).  Returns (n_pred, 10) bool: prediction is a true positive at each threshold.
    Per threshold (A.5): pairs with IoU >= thr and equal class, best IoU first; unique per prediction, then per GT."""
    n = pred_cls.shape[0]
    tp = np.zeros((n, IOU_THRESHOLDS.size), bool)
    if n == 0 or gt_cls.shape[0] == 0:
        return tp
    ok_cls = pred_cls[:, None] == gt_cls[None, :]
    for t, thr in enumerate(IOU_THRESHOLDS):
        pi, gi = np.nonzero((iou_pg >= thr) & ok_cls)
        if pi.size == 0:
            continue
        order = np.argsort(-iou_pg[pi, gi], kind="stable")
        pi, gi = pi[order], gi[order]
        keep = np.sort(np.unique(pi, return_index=True)[1])          # each prediction keeps its best pair ...
        pi, gi = pi[keep], gi[keep]
        keep = np.unique(gi, return_index=True)[1]                   # ... then each GT keeps its best remaining pair
        tp[pi[keep], t] = True
    return tp


def average_precision(recall: np.ndarray, precision: np.ndarray) -> float:
    r = np.concatenate(([0.0], recall, [1.0]))
    p = np.concatenate(([1.0], precision, [0.0]))
    p = np.maximum.accumulate(p[::-1])[::-1]
    x = np.linspace(0.0, 1.0, 101)
    y = np.interp(x, r, p)
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5))


def _smooth(y: np.ndarray, f: float = 0.05) -> np.ndarray:
    """Box filter of fraction f (upstream ``utils.metrics.smooth``; edges padded with the end values)."""
    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    yp = np.concatenate((pad * y[0], y, pad * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, gt_cls: np.ndarray) -> Dict[str, np.ndarray]:
    """tp (n,10); returns per-class AP (nc_present,10), precision / recall and the classes.  Precision / recall are read
    where upstream reads them: at ONE confidence for all classes, the argmax of the smoothed class-mean F1 curve over 1000
    confidence points (``ap_per_class`` upstream: ``i = smooth(f1_curve.mean(0), 0.1).argmax()``)."""
    order = np.argsort(-conf, kind="stable")
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_gt = np.unique(gt_cls.astype(np.int64), return_counts=True)
    ap = np.zeros((classes.size, tp.shape[1]))
    grid = np.linspace(0, 1, 1000)
    p_curve = np.zeros((classes.size, grid.size))
    r_curve = np.zeros((classes.size, grid.size))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        if not sel.any():
            continue
        tpc = np.cumsum(tp[sel], 0).astype(np.float64)
        fpc = np.cumsum(~tp[sel], 0).astype(np.float64)
        recall = tpc / (n_gt[ci] + 1e-16)
        precision = tpc / (tpc + fpc)
        for t in range(tp.shape[1]):
            ap[ci, t] = average_precision(recall[:, t], precision[:, t])
        r_curve[ci] = np.interp(-grid, -conf[sel], recall[:, 0], left=0)
        p_curve[ci] = np.interp(-grid, -conf[sel], precision[:, 0], left=1)
    f1 = 2 * p_curve * r_curve / (p_curve + r_curve + 1e-16)
    k = int(_smooth(f1.mean(0), 0.1).argmax()) if classes.size else 0
    return {"ap": ap, "precision": p_curve[:, k], "recall": r_curve[:, k], "classes": classes}


def summarize(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, gt_cls: np.ndarray) -> Tuple[float, float, float, float]:
    """(precision, recall, mAP50, mAP50-95), means over the classes present in the ground truth."""
    if gt_cls.size == 0:
        return 0.0, 0.0, 0.0, 0.0
    r = ap_per_class(tp, conf, pred_cls, gt_cls)
    return float(r["precision"].mean()), float(r["recall"].mean()), float(r["ap"][:, 0].mean()), float(r["ap"].mean())
