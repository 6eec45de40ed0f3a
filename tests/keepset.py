"""Margin-rule comparison of two end-to-end detection lists (SURVEY 8d "Tolerances").

Both sides ran forward + NMS on the same images.  The NMS itself is bit-exact for equal predictions
(test_engine_gpu.py::test_nms_exact_same_preds), so two keep-sets can differ only where the float noise of the
predictions crosses a decision of the greedy NMS.  SURVEY 8d: "class indices and NMS keep-set/order exact except for
detections whose score is within `m_conf` of `conf` or whose pairwise IoU is within `m_iou` of `iou` (these are
listed, not silently dropped)".  A detection kept by one side only must be EXPLAINED by one of:

  conf    its score on the other side is within m_conf of the confidence threshold;
  iou     on the other side it is suppressed by a kept box whose IoU with it is within m_iou of the IoU threshold;
  order   on the other side it is suppressed by a kept box whose score is within 2*m_conf of its own (the two swapped
          places in the sort: the same score noise as `conf`, seen between two overlapping candidates);
  cascade on the other side it is suppressed by a kept box that this side does not keep, and THAT difference is
          explained by one of the rules.

In every case the two sides' scores of the excepted anchor must agree within 2*m_conf.
Anything else is an unexplained difference and fails the test.  Every excepted detection is returned for printing.
Detections are identified by their anchor: a row's 32 mask coefficients are a bit copy of the prediction row.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def _xyxy(p: np.ndarray) -> np.ndarray:
    cx, cy, w, h = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    return np.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), 1).astype(np.float32)


def _iou(a: np.ndarray, b: np.ndarray) -> float:
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return float(inter / ua) if ua > 0 else 0.0


def anchors_of(dets: np.ndarray, preds: np.ndarray, nc: int = 1) -> List[int]:
    """dets (n, 6+nm) rows of one image, preds (A, 4+nc+nm) of the same side -> anchor index of every row."""
    table: Dict[bytes, int] = {}
    coefs = np.ascontiguousarray(preds[:, 4 + nc:])
    for a in range(coefs.shape[0] - 1, -1, -1):       # on a (never observed) duplicate the lowest anchor wins
        table[coefs[a].tobytes()] = a
    return [table[np.ascontiguousarray(r[6:]).tobytes()] for r in dets]


def compare_keepsets(keep_a: List[int], preds_a: np.ndarray, keep_b: List[int], preds_b: np.ndarray, conf: float,
                     iou: float, m_conf: float, m_iou: float, nc: int = 1) -> Tuple[List[Tuple[str, int, str]], List[Tuple[str, int]]]:
    """One image.  keep_*: kept anchors in output order; preds_*: (A, 4+nc+nm).  Returns (excepted, unexplained):
    excepted = [(side that keeps it, anchor, reason)], unexplained = [(side, anchor)]."""
    side = {"a": (set(keep_a), preds_a, keep_a), "b": (set(keep_b), preds_b, keep_b)}
    box = {k: _xyxy(v[1]) for k, v in side.items()}
    score = {k: v[1][:, 4:4 + nc].max(1) for k, v in side.items()}
    memo: Dict[Tuple[str, int], str] = {}

    def explain(kept_by: str, a: int, depth: int = 0) -> str:
        """Why does `kept_by` keep anchor a while the other side does not?  '' = no rule applies."""
        key = (kept_by, a)
        if key in memo:
            return memo[key]
        memo[key] = ""                                   # cycle guard
        other = "b" if kept_by == "a" else "a"
        s_o = float(score[other][a])
        if abs(float(score[kept_by][a]) - s_o) > 2 * m_conf:   # the two sides disagree about this anchor's score itself
            return ""
        if abs(s_o - conf) <= m_conf or s_o <= conf:     # not (safely) a candidate on the other side
            why = "conf" if abs(s_o - conf) <= m_conf else ""
            memo[key] = why
            return why
        # a candidate on the other side: something kept there suppresses it
        why = ""
        for r in side[other][2]:
            if r == a:
                continue
            v = _iou(box[other][a], box[other][r])
            if v <= iou - m_iou:
                continue
            if v <= iou + m_iou:
                why = "iou"
            elif abs(float(score[other][r]) - s_o) <= 2 * m_conf:
                why = "order"
            elif r not in side[kept_by][0] and depth < 8 and explain(other, r, depth + 1):
                why = "cascade"
            if why:
                break
        memo[key] = why
        return why

    excepted, unexplained = [], []
    for k, o in (("a", "b"), ("b", "a")):
        for a in side[k][2]:
            if a in side[o][0]:
                continue
            why = explain(k, a)
            (excepted if why else unexplained).append((k, a, why) if why else (k, a))
    return excepted, unexplained


def common_order_ok(keep_a: List[int], keep_b: List[int], score_b: np.ndarray, m_conf: float) -> bool:
    """The detections both sides keep come in the same order, except neighbours whose scores are within 2*m_conf."""
    common = set(keep_a) & set(keep_b)
    ia = [x for x in keep_a if x in common]
    ib = [x for x in keep_b if x in common]
    pos = {x: i for i, x in enumerate(ib)}
    for i, x in enumerate(ia):
        j = pos[x]
        if i != j and abs(float(score_b[x]) - float(score_b[ib[i]])) > 2 * m_conf:
            return False
    return True
