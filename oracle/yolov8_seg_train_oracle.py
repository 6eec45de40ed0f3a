"""CPU ORACLE for the TRAINING rows of the YOLOv8-seg path
--  TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as yolov8_seg_oracle.py).

PARITY STATUS: parity unpinned at the reference level (the loss / assigner / metric live in the
un-vendored ``ultralytics`` package reached from /root/reference/BscanBased/yolo_seg_train.py:12-19; the
reference holds no loss value, mAP or checkpoint for it).  Restated from the published algorithm
(TOOD task-aligned assignment, CIoU, Distribution Focal Loss, YOLACT-style prototype masks, COCO 101-point
AP) in plain PyTorch-CPU fp32 and pinned by the closed-form tests in tests/test_train_oracle.py.
Gradients of every function here come from PyTorch autograd, which is what the HIP backward kernels are
checked against.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from yolov8_seg_oracle import REG_MAX, make_anchors

EPS = 1e-7


# ----------------------------------------------------------------------------------------------
# IoU family (A.4): CIoU = IoU - (rho^2 / c^2 + v * alpha), alpha under no-grad
# ----------------------------------------------------------------------------------------------
def bbox_iou(b1: torch.Tensor, b2: torch.Tensor, ciou: bool = True) -> torch.Tensor:
    """xyxy boxes, broadcastable (..., 4) -> (..., 1)."""
    x1, y1, x2, y2 = b1.chunk(4, -1)
    X1, Y1, X2, Y2 = b2.chunk(4, -1)
    w1, h1 = x2 - x1, y2 - y1 + EPS
    w2, h2 = X2 - X1, Y2 - Y1 + EPS
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(0) * (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + EPS
    iou = inter / union
    if not ciou:
        return iou
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    ch = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    c2 = cw ** 2 + ch ** 2 + EPS
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + EPS))
    return iou - (rho2 / c2 + v * alpha)


# ----------------------------------------------------------------------------------------------
# TaskAlignedAssigner (A.4): topk 10, alpha 0.5, beta 6.0
# ----------------------------------------------------------------------------------------------
def task_aligned_assign(pd_scores: torch.Tensor, pd_bboxes: torch.Tensor, anc_points: torch.Tensor,
                        gt_labels: torch.Tensor, gt_bboxes: torch.Tensor, mask_gt: torch.Tensor,
                        topk: int = 10, alpha: float = 0.5, beta: float = 6.0, eps: float = 1e-9):
    """pd_scores (B,A,nc) sigmoid scores; pd_bboxes (B,A,4) xyxy pixels; anc_points (A,2) pixels;
    gt_labels (B,G,1) int; gt_bboxes (B,G,4) xyxy pixels; mask_gt (B,G,1) bool.
    Returns target_bboxes (B,A,4), target_scores (B,A,nc), fg_mask (B,A) bool, target_gt_idx (B,A)."""
    B, A, nc = pd_scores.shape
    G = gt_bboxes.shape[1]
    if G == 0:
        return (torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores), torch.zeros(B, A, dtype=torch.bool),
                torch.zeros(B, A, dtype=torch.long))
    lt = anc_points[None, None] - gt_bboxes[:, :, None, :2]           # (B,G,A,2)
    rb = gt_bboxes[:, :, None, 2:] - anc_points[None, None]
    mask_in_gts = torch.cat((lt, rb), -1).amin(-1) > eps               # centres strictly inside the GT
    mask = mask_in_gts & mask_gt.bool()                                # (B,G,A)
    cls_idx = gt_labels.long().squeeze(-1).clamp(0, nc - 1)            # (B,G)
    bbox_scores = torch.zeros(B, G, A)
    overlaps = torch.zeros(B, G, A)
    sc = pd_scores.permute(0, 2, 1)                                    # (B,nc,A)
    gathered = torch.gather(sc, 1, cls_idx[:, :, None].expand(B, G, A))
    bbox_scores[mask] = gathered[mask]
    iou = bbox_iou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :], ciou=True).squeeze(-1).clamp(0)
    overlaps[mask] = iou[mask]
    align = bbox_scores.pow(alpha) * overlaps.pow(beta)                # (B,G,A)
    # top-k anchors per GT by the alignment metric
    k = min(topk, A)
    topk_idx = torch.topk(align, k, dim=-1).indices                    # (B,G,k)
    mask_topk = torch.zeros(B, G, A, dtype=torch.bool)
    mask_topk.scatter_(2, topk_idx, True)
    mask_topk &= mask_gt.bool().expand(B, G, A)
    mask_pos = mask_topk & mask                                        # (B,G,A)
    # an anchor claimed by several GTs goes to the one with the highest overlap
    fg = mask_pos.sum(1)                                               # (B,A)
    if fg.max() > 1:
        multi = (fg > 1)[:, None, :].expand(B, G, A)
        best = overlaps.argmax(1)                                      # (B,A)
        is_best = torch.zeros(B, G, A, dtype=torch.bool).scatter_(1, best[:, None, :], True)
        mask_pos = torch.where(multi, is_best, mask_pos)
        fg = mask_pos.sum(1)
    target_gt_idx = mask_pos.float().argmax(1)                         # (B,A)
    fg_mask = fg > 0
    bidx = torch.arange(B)[:, None]
    target_labels = cls_idx[bidx, target_gt_idx]                       # (B,A)
    target_bboxes = gt_bboxes[bidx, target_gt_idx]                     # (B,A,4)
    target_scores = F.one_hot(target_labels, nc).float() * fg_mask[..., None]
    # normalise: score = align * max_overlap_of_gt / max_align_of_gt
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_over = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_over / (pos_align + eps)).amax(1)              # (B,A)
    target_scores = target_scores * norm[..., None]
    return target_bboxes, target_scores, fg_mask, target_gt_idx


# ----------------------------------------------------------------------------------------------
# v8SegmentationLoss (A.4): gains box 7.5, seg 7.5 (same hyper-parameter), cls 0.5, dfl 1.5
# ----------------------------------------------------------------------------------------------
def dist2bbox_xyxy(dist: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    lt, rb = dist.chunk(2, -1)
    return torch.cat((anchors - lt, anchors + rb), -1)


def bbox2dist(anchors: torch.Tensor, bbox: torch.Tensor, reg_max: int) -> torch.Tensor:
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchors - x1y1, x2y2 - anchors), -1).clamp_(0, reg_max - 0.01)


def dfl_loss(pred_dist: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """pred_dist (n*4, 16) logits; target (n,4) in [0, 15): two-hot cross entropy, mean over the 4 sides."""
    tl = target.long()
    tr = tl + 1
    wl = tr - target
    wr = 1 - wl
    return (F.cross_entropy(pred_dist, tl.view(-1), reduction="none").view(tl.shape) * wl
            + F.cross_entropy(pred_dist, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)


def crop_mask(masks: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    n, h, w = masks.shape
    x1, y1, x2, y2 = torch.chunk(boxes[:, :, None], 4, 1)
    r = torch.arange(w, dtype=x1.dtype)[None, None, :]
    c = torch.arange(h, dtype=x1.dtype)[None, :, None]
    return masks * ((r >= x1) * (r < x2) * (c >= y1) * (c < y2))


def single_mask_loss(gt_mask, pred_coef, proto, xyxy, area):
    """BCE(coef @ proto, gt) cropped to the box, mean over pixels, divided by the normalised box area."""
    pred = torch.einsum("in,nhw->ihw", pred_coef, proto)
    loss = F.binary_cross_entropy_with_logits(pred, gt_mask, reduction="none")
    return (crop_mask(loss, xyxy).mean(dim=(1, 2)) / area).sum()


def segmentation_loss(raw: Sequence[torch.Tensor], mc: torch.Tensor, proto: torch.Tensor, batch: Dict[str, torch.Tensor],
                      nc: int, imgsz: Tuple[int, int], gains=(7.5, 0.5, 1.5), overlap: bool = True):
    """raw: 3 maps (B,64+nc,h,w); mc (B,32,A); proto (B,32,mh,mw); batch: batch_idx (N,), cls (N,), bboxes (N,4)
    normalised xywh, masks (B,mh,mw) overlap-encoded (value = instance index + 1).
    Returns (loss * B, items[box, seg, cls, dfl])."""
    B = proto.shape[0]
    _, _, mh, mw = proto.shape
    no = raw[0].shape[1]
    x_cat = torch.cat([r.view(B, no, -1) for r in raw], 2)
    pred_distri, pred_scores = x_cat.split((REG_MAX * 4, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()            # (B,A,nc)
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()            # (B,A,64)
    pred_masks = mc.permute(0, 2, 1).contiguous()                      # (B,A,32)
    strides = [imgsz[0] // r.shape[2] for r in raw]
    anchor_points, stride_tensor = make_anchors([(r.shape[2], r.shape[3]) for r in raw], strides)
    # targets -> (B, G, 5) padded
    bi = batch["batch_idx"].long()
    counts = torch.bincount(bi, minlength=B)
    G = int(counts.max()) if bi.numel() else 0
    tg = torch.zeros(B, G, 5)
    for b in range(B):
        m = bi == b
        n = int(m.sum())
        if n:
            tg[b, :n, 0] = batch["cls"][m].float().view(-1)
            xywh = batch["bboxes"][m].float()
            scale = torch.tensor([imgsz[1], imgsz[0], imgsz[1], imgsz[0]], dtype=torch.float32)
            xy, wh = xywh[:, :2], xywh[:, 2:]
            tg[b, :n, 1:] = torch.cat((xy - wh / 2, xy + wh / 2), 1) * scale
    gt_labels, gt_bboxes = tg[..., :1], tg[..., 1:]
    mask_gt = gt_bboxes.sum(2, keepdim=True) > 0
    # decode
    proj = torch.arange(REG_MAX, dtype=torch.float32)
    pd = pred_distri.view(B, -1, 4, REG_MAX).softmax(3).matmul(proj)   # (B,A,4) grid units
    pred_bboxes = dist2bbox_xyxy(pd, anchor_points)                    # grid units
    tb, ts, fg, tgi = task_aligned_assign(pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor),
                                          anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt)
    tss = max(float(ts.sum()), 1.0)
    loss = torch.zeros(4)
    loss[2] = F.binary_cross_entropy_with_logits(pred_scores, ts, reduction="none").sum() / tss
    if fg.any():
        tbg = tb / stride_tensor
        w = ts.sum(-1)[fg][:, None]
        iou = bbox_iou(pred_bboxes[fg], tbg[fg], ciou=True)
        loss[0] = ((1.0 - iou) * w).sum() / tss
        tlrb = bbox2dist(anchor_points.expand(B, -1, -1)[fg], tbg[fg], REG_MAX - 1)  # clamp to 15 - 0.01
        loss[3] = (dfl_loss(pred_distri[fg].view(-1, REG_MAX), tlrb) * w).sum() / tss
        # masks
        masks = batch["masks"].float()
        txyxyn = tb / torch.tensor([imgsz[1], imgsz[0], imgsz[1], imgsz[0]], dtype=torch.float32)
        wh = txyxyn[..., 2:] - txyxyn[..., :2]
        marea = wh.prod(-1)
        mxyxy = txyxyn * torch.tensor([mw, mh, mw, mh], dtype=torch.float32)
        for b in range(B):
            f = fg[b]
            if f.any():
                idx = tgi[b][f]
                if overlap:
                    gt = (masks[b][None] == (idx + 1).view(-1, 1, 1)).float()
                else:
                    gt = masks[bi == b][idx]
                loss[1] = loss[1] + single_mask_loss(gt, pred_masks[b][f], proto[b], mxyxy[b][f], marea[b][f])
            else:
                loss[1] = loss[1] + (proto * 0).sum() + (pred_masks * 0).sum()
        loss[1] = loss[1] / fg.sum()
    else:
        loss[1] = loss[1] + (proto * 0).sum() + (pred_masks * 0).sum()
    box_gain, cls_gain, dfl_gain = gains
    loss = loss * torch.tensor([box_gain, box_gain, cls_gain, dfl_gain])
    return loss.sum() * B, loss.detach()


# ----------------------------------------------------------------------------------------------
# A17 / A.5: AP = area under the 101-point interpolated precision envelope
# ----------------------------------------------------------------------------------------------
def compute_ap(recall: np.ndarray, precision: np.ndarray) -> float:
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return float(trapz(np.interp(x, mrec, mpre), x))


def ap_per_class(tp: np.ndarray, conf: np.ndarray, pred_cls: np.ndarray, target_cls: np.ndarray):
    """tp (n, T) bool at T IoU thresholds.  Returns ap (n_classes, T) and the class ids."""
    order = np.argsort(-conf, kind="stable")
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    ap = np.zeros((len(classes), tp.shape[1]))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_gt = int((target_cls == c).sum())
        if sel.sum() == 0 or n_gt == 0:
            continue
        tpc = tp[sel].cumsum(0)
        fpc = (1 - tp[sel]).cumsum(0)
        recall = tpc / (n_gt + 1e-16)
        precision = tpc / (tpc + fpc)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    return ap, classes


def match_predictions(pred_cls: np.ndarray, true_cls: np.ndarray, iou: np.ndarray,
                      iouv: np.ndarray = np.linspace(0.5, 0.95, 10)) -> np.ndarray:
    """iou (n_gt, n_pred).  Greedy matching per threshold: pairs sorted by IoU, unique per prediction then per GT."""
    correct = np.zeros((pred_cls.shape[0], iouv.shape[0]), bool)
    same = true_cls[:, None] == pred_cls[None, :]
    iou = iou * same
    for i, thr in enumerate(iouv):
        g, p = np.nonzero(iou >= thr)
        if g.size:
            m = np.stack((g, p, iou[g, p]), 1)
            m = m[np.argsort(-m[:, 2], kind="stable")]
            m = m[np.unique(m[:, 1], return_index=True)[1]]
            m = m[np.argsort(-m[:, 2], kind="stable")]
            m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), i] = True
    return correct


# ----------------------------------------------------------------------------------------------
# A14: schedule pieces
# ----------------------------------------------------------------------------------------------
def lr_lambda(epoch: int, epochs: int, lrf: float = 0.01) -> float:
    return max(1 - epoch / epochs, 0) * (1.0 - lrf) + lrf


def ema_decay(updates: int, decay: float = 0.9999, tau: float = 2000.0) -> float:
    return decay * (1 - math.exp(-updates / tau))
