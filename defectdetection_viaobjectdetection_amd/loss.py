"""Segmentation training loss on the head outputs of the HIP engine.

Stands where ``v8SegmentationLoss`` + ``TaskAlignedAssigner`` + ``BboxLoss`` stand upstream (reached from
/root/reference/BscanBased/yolo_seg_train.py:12).  Inputs are the engine's train-mode outputs -- raw head maps
``(B, A, 64+nc+32)`` fp32 and prototypes ``(B, mh, mw, 32)`` NHWC -- so the autograd graph of this module is only
the loss itself (a few MB of irregular gathers); its gradients w.r.t. those two tensors are what
``TrainEngine.backward`` consumes.  Plain PyTorch tensor ops, device agnostic; no convolution happens here.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

REG_MAX = 16
NM = 32
_EPS = 1e-7


def anchor_grid(imgsz: Tuple[int, int], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Cell centres (x+0.5, y+0.5) in grid units, row-major per level P3,P4,P5, and the stride per anchor."""
    pts, strides = [], []
    for s in (8, 16, 32):
        h, w = imgsz[0] // s, imgsz[1] // s
        ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32) + 0.5,
                                torch.arange(w, device=device, dtype=torch.float32) + 0.5, indexing="ij")
        pts.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), 1))
        strides.append(torch.full((h * w, 1), float(s), device=device))
    return torch.cat(pts), torch.cat(strides)


def ciou(b1: torch.Tensor, b2: torch.Tensor, complete: bool = True) -> torch.Tensor:
    """Complete-IoU of xyxy boxes (broadcasting over leading dims); returns (...,)."""
    ax1, ay1, ax2, ay2 = b1.unbind(-1)
    bx1, by1, bx2, by2 = b2.unbind(-1)
    aw, ah = ax2 - ax1, ay2 - ay1 + _EPS
    bw, bh = bx2 - bx1, by2 - by1 + _EPS
    iw = (torch.minimum(ax2, bx2) - torch.maximum(ax1, bx1)).clamp_min(0)
    ih = (torch.minimum(ay2, by2) - torch.maximum(ay1, by1)).clamp_min(0)
    inter = iw * ih
    iou = inter / (aw * ah + bw * bh - inter + _EPS)
    if not complete:
        return iou
    cw = torch.maximum(ax2, bx2) - torch.minimum(ax1, bx1)
    chh = torch.maximum(ay2, by2) - torch.minimum(ay1, by1)
    diag2 = cw * cw + chh * chh + _EPS
    centre2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) * 0.25
    v = (4.0 / math.pi ** 2) * (torch.atan(bw / bh) - torch.atan(aw / ah)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1.0 + _EPS))
    return iou - (centre2 / diag2 + v * alpha)


@torch.no_grad()
def assign_targets(scores: torch.Tensor, boxes: torch.Tensor, anchors_px: torch.Tensor, gt_cls: torch.Tensor,
                   gt_boxes: torch.Tensor, gt_valid: torch.Tensor, topk: int = 10, alpha: float = 0.5,
                   beta: float = 6.0, eps: float = 1e-9):
    """Task-aligned assignment.  scores (B,A,nc) in [0,1]; boxes (B,A,4) xyxy px; anchors_px (A,2);
    gt_cls (B,G) long; gt_boxes (B,G,4) xyxy px; gt_valid (B,G) bool.
    Returns target boxes (B,A,4), target scores (B,A,nc), foreground mask (B,A), assigned GT index (B,A)."""
    B, A, nc = scores.shape
    G = gt_boxes.shape[1]
    if G == 0:
        z = torch.zeros(B, A, dtype=torch.long, device=scores.device)
        return torch.zeros_like(boxes), torch.zeros_like(scores), z.bool(), z
    d = torch.cat((anchors_px[None, None] - gt_boxes[:, :, None, :2], gt_boxes[:, :, None, 2:] - anchors_px[None, None]), -1)
    cand = (d.amin(-1) > eps) & gt_valid[:, :, None]                              # (B,G,A) centre strictly inside
    cls_score = scores.transpose(1, 2).gather(1, gt_cls.clamp(0, nc - 1)[:, :, None].expand(B, G, A))
    overlap = ciou(gt_boxes[:, :, None, :], boxes[:, None, :, :]).clamp_min(0) * cand
    metric = (cls_score * cand).pow(alpha) * overlap.pow(beta)
    top = torch.zeros_like(cand)
    top.scatter_(2, metric.topk(min(topk, A), dim=2).indices, True)
    pos = top & cand
    claims = pos.sum(1)                                                          # several GTs -> highest overlap wins
    winner = torch.zeros_like(pos).scatter_(1, overlap.argmax(1, keepdim=True), True)
    pos = torch.where((claims > 1)[:, None, :], winner, pos)
    fg = pos.any(1)
    gt_idx = pos.float().argmax(1)
    bi = torch.arange(B, device=scores.device)[:, None]
    t_boxes = gt_boxes[bi, gt_idx]
    t_cls = gt_cls[bi, gt_idx].clamp(0, nc - 1)
    metric = metric * pos
    norm = (metric * (overlap * pos).amax(2, keepdim=True) / (metric.amax(2, keepdim=True) + eps)).amax(1)
    t_scores = F.one_hot(t_cls, nc).to(scores.dtype) * (fg * norm)[..., None]
    return t_boxes, t_scores, fg, gt_idx


def segmentation_loss(raw: torch.Tensor, protos: torch.Tensor, batch: Dict[str, torch.Tensor], nc: int,
                      imgsz: Tuple[int, int], box_gain: float = 7.5, cls_gain: float = 0.5, dfl_gain: float = 1.5):
    """raw (B,A,64+nc+32); protos (B,mh,mw,32).  batch: batch_idx (N,), cls (N,), bboxes (N,4) xywh normalised to the
    network input, masks (B,mh,mw) overlap-encoded (pixel value = 1 + index of the instance within its image,
    instances sorted by area descending).  Returns (loss * B, detached items [box, seg, cls, dfl])."""
    dev = raw.device
    B, A, _ = raw.shape
    mh, mw = protos.shape[1:3]
    logits_box, logits_cls, coefs = raw.split((4 * REG_MAX, nc, NM), 2)
    anchors, strides = anchor_grid(imgsz, dev)
    bins = torch.arange(REG_MAX, device=dev, dtype=torch.float32)
    # expectation over the 16 bins as multiply + reduce: `softmax @ bins` runs as a (B*A*4) x 16 rocBLAS gemv, 1.4 ms
    ltrb = (logits_box.view(B, A, 4, REG_MAX).softmax(3) * bins).sum(3)
    pred_boxes = torch.cat((anchors - ltrb[..., :2], anchors + ltrb[..., 2:]), -1)   # grid units
    # padded GT tensors
    bidx = batch["batch_idx"].to(dev).long()
    n_per = torch.bincount(bidx, minlength=B) if bidx.numel() else torch.zeros(B, dtype=torch.long, device=dev)
    G = int(n_per.max()) if bidx.numel() else 0
    gt_cls = torch.zeros(B, G, dtype=torch.long, device=dev)
    gt_boxes = torch.zeros(B, G, 4, device=dev)
    gt_valid = torch.zeros(B, G, dtype=torch.bool, device=dev)
    if G:
        order = torch.argsort(bidx, stable=True)
        slot = torch.arange(bidx.numel(), device=dev) - torch.cumsum(n_per, 0)[bidx[order]] + n_per[bidx[order]]
        xywh = batch["bboxes"].to(dev).float()[order]
        scale = torch.tensor([imgsz[1], imgsz[0], imgsz[1], imgsz[0]], device=dev, dtype=torch.float32)
        xyxy = torch.cat((xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2), 1) * scale
        gt_boxes[bidx[order], slot] = xyxy
        gt_cls[bidx[order], slot] = batch["cls"].to(dev).long().view(-1)[order]
        gt_valid[bidx[order], slot] = xyxy.sum(1) > 0
    t_boxes, t_scores, fg, gt_idx = assign_targets(logits_cls.detach().sigmoid(), pred_boxes.detach() * strides,
                                                   anchors * strides, gt_cls, gt_boxes, gt_valid)
    # Everything below runs over ALL anchors / a fixed number of slots per image with zero weights for the background,
    # so the step has no data-dependent shapes and no host synchronisation after the one that sized the GT padding.
    denom = t_scores.sum().clamp_min(1.0)
    w = t_scores.sum(-1)                                                            # (B,A), 0 off the foreground
    loss_cls = F.binary_cross_entropy_with_logits(logits_cls, t_scores, reduction="sum") / denom
    if G == 0:
        zero = (protos * 0).sum() + (coefs * 0).sum() + (logits_box * 0).sum()
        items = torch.stack((zero, zero, loss_cls, zero))
    else:
        tb = t_boxes / strides
        loss_box = ((1.0 - ciou(pred_boxes, tb)) * w).sum() / denom
        dist = torch.cat((anchors - tb[..., :2], tb[..., 2:] - anchors), -1).clamp(0, REG_MAX - 1 - 0.01)   # (B,A,4)
        lo = dist.long()
        logp = logits_box.view(B, A, 4, REG_MAX).log_softmax(3)
        ce_lo = -logp.gather(3, lo[..., None]).squeeze(3)
        ce_hi = -logp.gather(3, lo[..., None] + 1).squeeze(3)
        loss_dfl = (((ce_lo * (lo + 1 - dist) + ce_hi * (dist - lo)).mean(2)) * w).sum() / denom
        # masks: every GT claims at most 10 anchors, so K = 10 G slots per image hold all foreground anchors.
        # BCE(coef . proto, gt mask of the assigned instance) inside the target box, mean over the map, divided by
        # the normalised box area; one batched GEMM (B,K,32) x (B,32,mh*mw).
        K = min(10 * G, A)
        val, ai = fg.float().topk(K, dim=1)
        valid = val > 0                                                              # (B,K)
        wh = torch.tensor([imgsz[1], imgsz[0], imgsz[1], imgsz[0]], device=dev, dtype=torch.float32)
        nb = t_boxes.gather(1, ai[..., None].expand(B, K, 4)) / wh
        area = ((nb[..., 2] - nb[..., 0]) * (nb[..., 3] - nb[..., 1])).masked_fill(~valid, 1.0)
        mb = (nb * torch.tensor([mw, mh, mw, mh], device=dev, dtype=torch.float32))[..., None]            # (B,K,4,1)
        cols = torch.arange(mw, device=dev, dtype=torch.float32).repeat(mh)[None, None, :]
        rows = torch.arange(mh, device=dev, dtype=torch.float32).repeat_interleave(mw)[None, None, :]
        inside = (cols >= mb[:, :, 0]) & (cols < mb[:, :, 2]) & (rows >= mb[:, :, 1]) & (rows < mb[:, :, 3])
        ck = coefs.gather(1, ai[..., None].expand(B, K, NM))
        pred = torch.bmm(ck, protos.float().reshape(B, mh * mw, NM).transpose(1, 2))                    # (B,K,HW)
        inst = (gt_idx.gather(1, ai) + 1)[..., None]
        gt = (batch["masks"].to(dev).reshape(B, 1, mh * mw) == inst).to(pred.dtype)
        bce = F.binary_cross_entropy_with_logits(pred, gt, reduction="none")
        per_slot = (bce * inside).mean(2) / area
        loss_seg = (per_slot * valid).sum() / fg.sum().clamp_min(1)
        items = torch.stack((loss_box, loss_seg, loss_cls, loss_dfl))
    gains = torch.tensor([box_gain, box_gain, cls_gain, dfl_gain], device=dev)
    items = items * gains
    return items.sum() * B, items.detach()
