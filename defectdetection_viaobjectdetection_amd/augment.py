"""Training augmentation: parameters and label geometry on the host, pixels on the GPU (SURVEY.md A14 defaults, N2).

Upstream's defaults for the reference's train call (/root/reference/BscanBased/yolo_seg_train.py:12): mosaic 1.0
(off for the last `close_mosaic` = 10 epochs), RandomPerspective(degrees 0, translate 0.1, scale 0.5, shear 0,
perspective 0), HSV (0.015, 0.7, 0.4), fliplr 0.5.  The letterboxed uint8 cache of the split lives in HBM; one launch
of `m355_augment` (csrc/augment.hip) composes the mosaic canvas, warps it bilinearly with border 114, applies the HSV
gains and the flip and writes the network input batch.  The polygons go through the same forward matrix here, are
clipped to the image (Sutherland-Hodgman) and filtered like upstream's `box_candidates` (>= 2 px wide and high).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from ._capi import AugParams, check, lib
from .dataset import SegDataset, overlap_mask

HYP = dict(mosaic=1.0, scale=0.5, translate=0.1, hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, fliplr=0.5)


def clip_polygon(poly: np.ndarray, w: float, h: float) -> np.ndarray:
    """Sutherland-Hodgman clip of an (n,2) polygon to [0,w] x [0,h]; returns (m,2), m may be 0."""
    def clip_edge(pts, axis, bound, keep_less):
        out = []
        n = len(pts)
        for i in range(n):
            a, b = pts[i], pts[(i + 1) % n]
            ina = a[axis] <= bound if keep_less else a[axis] >= bound
            inb = b[axis] <= bound if keep_less else b[axis] >= bound
            if ina:
                out.append(a)
            if ina != inb:
                t = (bound - a[axis]) / (b[axis] - a[axis])
                out.append(a + t * (b - a))
        return out
    pts = [p for p in np.asarray(poly, np.float64)]
    for axis, bound, less in ((0, 0.0, False), (0, float(w), True), (1, 0.0, False), (1, float(h), True)):
        if not pts:
            break
        pts = clip_edge(pts, axis, bound, less)
    return np.asarray(pts, np.float64).reshape(-1, 2)


def random_affine(rng: np.random.Generator, out_hw: Tuple[int, int], canvas_hw: Tuple[int, int], scale: float, translate: float):
    """Forward 3x3 matrix canvas -> output: centre the canvas, scale by U(1-scale, 1+scale), move the centre to
    U(0.5 - translate, 0.5 + translate) of the output."""
    H, W = out_hw
    cc = np.eye(3)
    cc[0, 2], cc[1, 2] = -canvas_hw[1] / 2.0, -canvas_hw[0] / 2.0
    s = rng.uniform(1.0 - scale, 1.0 + scale)
    r = np.diag([s, s, 1.0])
    t = np.eye(3)
    t[0, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * W
    t[1, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * H
    return t @ r @ cc


class Augmenter:
    def __init__(self, ds: SegDataset, device: torch.device, seed: int = 0, **hyp):
        unknown = [k for k in hyp if k not in HYP]
        if unknown:
            raise TypeError(f"unknown augmentation options {unknown}")
        self.ds, self.dev = ds, device
        self.hyp = {**HYP, **hyp}
        self.rng = np.random.default_rng(seed)
        self.cache = torch.from_numpy(ds.images).to(device)          # uint8 (N,H,W,3), resident for the whole run

    def plan(self, indices: Sequence[int], mosaic_on: bool = True) -> List[dict]:
        """Random parameters + transformed labels of one batch (host)."""
        H, W = self.ds.imgsz
        hy, rng, n = self.hyp, self.rng, len(self.ds)
        out = []
        for i in indices:
            mosaic = bool(mosaic_on and rng.random() < hy["mosaic"])
            if mosaic:
                src = [int(i)] + [int(v) for v in rng.integers(0, n, 3)]
                xc, yc = int(rng.uniform(0.5 * W, 1.5 * W)), int(rng.uniform(0.5 * H, 1.5 * H))
                offs = [(xc - W, yc - H), (xc, yc - H), (xc - W, yc), (xc, yc)]
                m = random_affine(rng, (H, W), (2 * H, 2 * W), hy["scale"], hy["translate"])
            else:
                src, xc, yc, offs = [int(i)] * 4, 0, 0, [(0, 0)]
                m = random_affine(rng, (H, W), (H, W), hy["scale"], hy["translate"])
            flip = bool(rng.random() < hy["fliplr"])
            gains = rng.uniform(-1, 1, 3) * np.array([hy["hsv_h"], hy["hsv_s"], hy["hsv_v"]]) + 1.0
            inst = []
            for k, (ox, oy) in enumerate(offs):
                for c, poly in self.ds.labels[src[k]]:
                    if mosaic:                                       # the part of the source visible on the canvas
                        poly = clip_polygon(poly + np.array([ox, oy], np.float64), 2 * W, 2 * H)
                        if len(poly) < 3:
                            continue
                    q = poly @ m[:2, :2].T + m[:2, 2]
                    q = clip_polygon(q, W, H)
                    if len(q) < 3:
                        continue
                    if flip:
                        q = np.stack((W - q[:, 0], q[:, 1]), 1)
                    bw, bh = q[:, 0].max() - q[:, 0].min(), q[:, 1].max() - q[:, 1].min()
                    if bw < 2 or bh < 2:
                        continue
                    inst.append((c, q))
            out.append(dict(src=src, xc=xc, yc=yc, m=m, flip=flip, gains=gains, mosaic=mosaic, inst=inst))
        return out

    def render(self, plans: List[dict]) -> torch.Tensor:
        """One m355_augment launch: uint8 (B,H,W,3) on the device."""
        H, W = self.ds.imgsz
        B = len(plans)
        arr = (AugParams * B)()
        for b, p in enumerate(plans):
            minv = np.linalg.inv(p["m"])
            a = arr[b]
            for k in range(4):
                a.src[k] = p["src"][k]
            a.xc, a.yc = float(p["xc"]), float(p["yc"])
            for k, v in enumerate(minv[:2].reshape(-1)):
                a.minv[k] = float(v)
            a.hgain, a.sgain, a.vgain = (float(g) for g in p["gains"])
            a.flip, a.mosaic = int(p["flip"]), int(p["mosaic"])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.dev)
        check(lib.m355_augment(C.c_void_p(self.cache.data_ptr()), C.c_void_p(raw.data_ptr()), C.c_void_p(out.data_ptr()), B, H, W,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        out._keepalive = raw
        return out

    def batch(self, indices: Sequence[int], mosaic_on: bool = True) -> Dict:
        """Like SegDataset.batch, with `img` already on the device."""
        H, W = self.ds.imgsz
        plans = self.plan(indices, mosaic_on)
        imgs = self.render(plans)
        bidx, cls, boxes = [], [], []
        masks = np.zeros((len(plans), H // 4, W // 4), np.uint8)
        if any(len(p["inst"]) > 255 for p in plans):                 # (a mosaic of four crowded images)
            masks = masks.astype(np.int32)
        for b, p in enumerate(plans):
            polys = [q for _, q in p["inst"]]
            if not polys:
                continue
            masks[b], order = overlap_mask(polys, (H, W))
            for j in order:
                q = polys[j]
                x1, y1, x2, y2 = q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max()
                bidx.append(b)
                cls.append(p["inst"][j][0])
                boxes.append([(x1 + x2) / 2 / W, (y1 + y2) / 2 / H, (x2 - x1) / W, (y2 - y1) / H])
        return {"img": imgs, "batch_idx": np.asarray(bidx, np.float32), "cls": np.asarray(cls, np.float32),
                "bboxes": np.asarray(boxes, np.float32).reshape(-1, 4), "masks": masks, "plans": plans}
