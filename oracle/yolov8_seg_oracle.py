"""CPU ORACLE for the YOLOv8-seg hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and there only as the checker / the timed CPU baseline.  The product package
(``defectdetection_viaobjectdetection_amd`` and the ``ultralytics`` shim) never imports it.

PARITY STATUS: **parity unpinned at the reference level.**  The arithmetic of the path the reference
scripts exercise (``/root/reference/BscanBased/yolo8_seg_predict.py:5-9``,
``/root/reference/BscanBased/yolo_seg_train.py:7-19``) lives in the third-party ``ultralytics``
package (PyPI, AGPL-3.0, 8.x; un-vendored, un-pinned, not installed, not installable offline) and
the reference holds no test, golden vector or saved prediction for it.
This file therefore restates the *published* YOLOv8-seg algorithm (SURVEY.md section 8a rows A3-A12
and Appendix A) in plain PyTorch-CPU fp32 / numpy and is pinned by
  (i)   exact parameter counts (nc=80: 3 409 968 / 11 821 056 / 27 285 968 for n/s/m -- the
        published 3.4 / 11.8 / 27.3 M; nc=1: 3 263 811 / 11 790 483 / 27 240 227),
  (ii)  output shapes (B, 4+nc+32, 8400) and (B, 32, 160, 160) at 640x640,
  (iii) closed-form known-answer tests per stage (tests/test_oracle_known_answers.py),
  (iv)  the reference's own committed inputs (BscanBased/yolo/*.png, annotations.json excerpts
        under tests/golden/).

Each function cites the SURVEY row (and through it the reference call site) it restates.
Written from the behavioural spec, not from upstream source.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# A5: graph constants (yolov8-seg.yaml [U]; validated by param counts)
# ----------------------------------------------------------------------------------------------
SCALES = {  # depth, width, max_channels
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}
REG_MAX = 16
NM = 32  # number of mask prototypes


def make_divisible(x: float, d: int) -> int:
    return int(math.ceil(x / d) * d)


def autopad(k: int) -> int:
    return k // 2


# ----------------------------------------------------------------------------------------------
# A4: Conv = Conv2d(bias=False) + BatchNorm2d(eps=1e-3, momentum=0.03) + SiLU
# ----------------------------------------------------------------------------------------------
class Conv(nn.Module):
    def __init__(self, c1: int, c2: int, k: int = 1, s: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k), bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)

    def forward(self, x):
        return F.silu(self.bn(self.conv(x)))


# A6: Bottleneck / C2f
class Bottleneck(nn.Module):
    def __init__(self, c1: int, c2: int, shortcut: bool):
        super().__init__()
        c_ = int(c2 * 1.0)  # e = 1.0 inside C2f
        self.cv1 = Conv(c1, c_, 3, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    def __init__(self, c1: int, c2: int, n: int, shortcut: bool):
        super().__init__()
        self.c = int(c2 * 0.5)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


# A7: SPPF
class SPPF(nn.Module):
    def __init__(self, c1: int, c2: int, k: int = 5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k

    def forward(self, x):
        a = self.cv1(x)
        p1 = F.max_pool2d(a, self.k, 1, self.k // 2)
        p2 = F.max_pool2d(p1, self.k, 1, self.k // 2)
        p3 = F.max_pool2d(p2, self.k, 1, self.k // 2)
        return self.cv2(torch.cat((a, p1, p2, p3), 1))


# A10: Proto
class Proto(nn.Module):
    def __init__(self, c1: int, c_: int, c2: int):
        super().__init__()
        self.cv1 = Conv(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = Conv(c_, c_, 3)
        self.cv3 = Conv(c_, c2, 1)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


class DFL(nn.Module):
    def __init__(self, c1: int = REG_MAX):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float32).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):  # (B, 4*c1, A) -> (B, 4, A)
        b, _, a = x.shape
        return self.conv(x.view(b, 4, self.c1, a).transpose(2, 1).softmax(1)).view(b, 4, a)


def make_anchors(shapes: Sequence[Tuple[int, int]], strides: Sequence[int]):
    """A9 / A.4: cell centres (+0.5), row-major (y outer), levels concatenated P3,P4,P5."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + 0.5
        sy = torch.arange(h, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


# A9 + A10: Segment head (Detect + Proto + mask-coefficient branch)
class Segment(nn.Module):
    def __init__(self, nc: int, nm: int, npr: int, ch: Sequence[int]):
        super().__init__()
        self.nc, self.nm, self.npr = nc, nm, npr
        self.nl = len(ch)
        self.no = nc + REG_MAX * 4
        self.stride = torch.tensor([8.0, 16.0, 32.0])
        c2 = max(16, ch[0] // 4, REG_MAX * 4)
        c3 = max(ch[0], min(nc, 100))
        c4 = max(ch[0] // 4, nm)
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * REG_MAX, 1)) for x in ch)
        self.cv3 = nn.ModuleList(
            nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
        self.cv4 = nn.ModuleList(
            nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch)
        self.dfl = DFL(REG_MAX)
        self.proto = Proto(ch[0], npr, nm)

    def bias_init(self, imgsz: int = 640):
        """A.1: head bias init."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (imgsz / float(s)) ** 2)

    def forward_raw(self, feats: List[torch.Tensor]):
        """A13: raw per-level maps (B, 64+nc, h, w), coefficients (B, 32, A), protos."""
        p = self.proto(feats[0])
        bs = p.shape[0]
        mc = torch.cat([self.cv4[i](feats[i]).view(bs, self.nm, -1) for i in range(self.nl)], 2)
        raw = [torch.cat((self.cv2[i](feats[i]), self.cv3[i](feats[i])), 1) for i in range(self.nl)]
        return raw, mc, p

    def forward(self, feats: List[torch.Tensor]):
        raw, mc, p = self.forward_raw(feats)
        bs = p.shape[0]
        shapes = [(r.shape[2], r.shape[3]) for r in raw]
        x_cat = torch.cat([r.view(bs, self.no, -1) for r in raw], 2)
        box, cls = x_cat.split((REG_MAX * 4, self.nc), 1)
        anchors, strides = make_anchors(shapes, [int(s) for s in self.stride])
        dist = self.dfl(box)  # (B, 4, A): l, t, r, b in grid units
        lt, rb = dist.chunk(2, 1)
        a = anchors.t().unsqueeze(0)
        x1y1 = a - lt
        x2y2 = a + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides.t().unsqueeze(0)
        y = torch.cat((dbox, cls.sigmoid()), 1)
        return torch.cat((y, mc), 1), p


class SegmentationModel(nn.Module):
    """A5: the 23-entry yolov8-seg graph; layer i is ``self.model[i]`` (A.1 naming)."""

    def __init__(self, scale: str = "s", nc: int = 1):
        super().__init__()
        depth, width, maxc = SCALES[scale]

        def ch(c):
            return make_divisible(min(c, maxc) * width, 8)

        def rep(n):
            return max(round(n * depth), 1) if n > 1 else n

        c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
        m: List[nn.Module] = [
            Conv(3, c64, 3, 2),                              # 0
            Conv(c64, c128, 3, 2),                           # 1
            C2f(c128, c128, rep(3), True),                   # 2
            Conv(c128, c256, 3, 2),                          # 3
            C2f(c256, c256, rep(6), True),                   # 4
            Conv(c256, c512, 3, 2),                          # 5
            C2f(c512, c512, rep(6), True),                   # 6
            Conv(c512, c1024, 3, 2),                         # 7
            C2f(c1024, c1024, rep(3), True),                 # 8
            SPPF(c1024, c1024, 5),                           # 9
            nn.Upsample(scale_factor=2, mode="nearest"),     # 10
            nn.Identity(),                                   # 11 Concat[-1, 6]
            C2f(c1024 + c512, c512, rep(3), False),          # 12
            nn.Upsample(scale_factor=2, mode="nearest"),     # 13
            nn.Identity(),                                   # 14 Concat[-1, 4]
            C2f(c512 + c256, c256, rep(3), False),           # 15
            Conv(c256, c256, 3, 2),                          # 16
            nn.Identity(),                                   # 17 Concat[-1, 12]
            C2f(c256 + c512, c512, rep(3), False),           # 18
            Conv(c512, c512, 3, 2),                          # 19
            nn.Identity(),                                   # 20 Concat[-1, 9]
            C2f(c512 + c1024, c1024, rep(3), False),         # 21
            Segment(nc, NM, ch(256), (c256, c512, c1024)),   # 22
        ]
        self.model = nn.ModuleList(m)
        self.nc = nc
        self.scale = scale
        self.model[22].bias_init(640)

    def features(self, x):
        m = self.model
        x0 = m[0](x)
        x1 = m[1](x0)
        x2 = m[2](x1)
        x3 = m[3](x2)
        x4 = m[4](x3)
        x5 = m[5](x4)
        x6 = m[6](x5)
        x7 = m[7](x6)
        x8 = m[8](x7)
        x9 = m[9](x8)
        x12 = m[12](torch.cat((m[10](x9), x6), 1))
        x15 = m[15](torch.cat((m[13](x12), x4), 1))
        x18 = m[18](torch.cat((m[16](x15), x12), 1))
        x21 = m[21](torch.cat((m[19](x18), x9), 1))
        return [x15, x18, x21]

    def forward(self, x):
        """Inference forward: preds (B, 4+nc+32, A), protos (B, 32, H/4, W/4)."""
        return self.model[22](self.features(x))

    def forward_raw(self, x):
        return self.model[22].forward_raw(self.features(x))


def count_parameters(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())


def conv_macs_per_image(scale: str, nc: int, imgsz: int = 640) -> int:
    """Conv-only MACs per image (BN folded, ConvT counted Cin*Cout per output pixel) -- SURVEY 8d."""
    model = SegmentationModel(scale, nc).eval()
    macs = 0
    hooks = []

    def hook(mod, inp, out):
        nonlocal macs
        if isinstance(mod, nn.ConvTranspose2d):
            macs += out.shape[2] * out.shape[3] * mod.in_channels * mod.out_channels
        elif mod.weight.requires_grad or True:
            k = mod.kernel_size[0] * mod.kernel_size[1]
            macs += out.shape[2] * out.shape[3] * mod.out_channels * (mod.in_channels // mod.groups) * k

    for mod in model.modules():
        if isinstance(mod, (nn.Conv2d, nn.ConvTranspose2d)) and not isinstance(mod, type(None)):
            if mod is model.model[22].dfl.conv:
                continue
            hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        model(torch.zeros(1, 3, imgsz, imgsz))
    for h in hooks:
        h.remove()
    return macs


# ----------------------------------------------------------------------------------------------
# A3 / A.2: LetterBox pre-processing (PIL-free, numpy; emulates cv2.INTER_LINEAR on uint8)
# ----------------------------------------------------------------------------------------------
def resize_bilinear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Half-pixel-centre bilinear resize of a uint8 HxWxC image, edges replicated, result rounded to
    nearest (A.2; cv2's fixed-point rounding may differ by 1 LSB -- H7)."""
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()
    ys = (np.arange(new_h, dtype=np.float64) + 0.5) * (h / new_h) - 0.5
    xs = (np.arange(new_w, dtype=np.float64) + 0.5) * (w / new_w) - 0.5
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    im = img.astype(np.float64)
    top = im[y0c][:, x0c] * (1 - fx) + im[y0c][:, x1c] * fx
    bot = im[y1c][:, x0c] * (1 - fx) + im[y1c][:, x1c] * fx
    out = top * (1 - fy) + bot * fy
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_shape: Tuple[int, int] = (640, 640), auto: bool = False,
              stride: int = 32, color: int = 114):
    """A.2.  img: HxWx3 uint8.  Returns (padded image, ratio, (left, top) padding)."""
    h, w = img.shape[:2]
    r = min(new_shape[0] / h, new_shape[1] / w)
    unpad_w, unpad_h = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape[1] - unpad_w, new_shape[0] - unpad_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    if (h, w) != (unpad_h, unpad_w):
        img = resize_bilinear_u8(img, unpad_h, unpad_w)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.full((unpad_h + top + bottom, unpad_w + left + right, img.shape[2]), color, np.uint8)
    out[top:top + unpad_h, left:left + unpad_w] = img
    return out, r, (left, top)


def preprocess(imgs_u8_hwc: Sequence[np.ndarray]) -> torch.Tensor:
    """A.2 tail: stack, HWC->CHW, /255 -> float32 (B,3,H,W).  Channel order is whatever the caller
    gives (gray B-scans replicate one channel, so BGR<->RGB is the identity for the fixtures)."""
    x = np.stack(imgs_u8_hwc).transpose(0, 3, 1, 2)
    return torch.from_numpy(np.ascontiguousarray(x)).float() / 255.0


# ----------------------------------------------------------------------------------------------
# A11: non_max_suppression  (numpy float32; op order mirrored by the HIP kernel for bit-exactness)
# ----------------------------------------------------------------------------------------------
MAX_WH = 7680.0
MAX_NMS = 30000


def box_iou_f32(b: np.ndarray, others: np.ndarray) -> np.ndarray:
    """IoU of one xyxy box against N boxes, float32, torchvision-nms op order:
    inter / (area_a + area_b - inter)."""
    b = b.astype(np.float32)
    o = others.astype(np.float32)
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    area_o = (o[:, 2] - o[:, 0]) * (o[:, 3] - o[:, 1])
    w = np.maximum(np.float32(0), np.minimum(b[2], o[:, 2]) - np.maximum(b[0], o[:, 0]))
    h = np.maximum(np.float32(0), np.minimum(b[3], o[:, 3]) - np.maximum(b[1], o[:, 1]))
    inter = (w * h).astype(np.float32)
    union = ((area_b + area_o).astype(np.float32) - inter).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(np.float32)


def nms_greedy(boxes: np.ndarray, iou_thr: float, max_keep: int) -> np.ndarray:
    """Greedy NMS over boxes already sorted by score (desc).  Keep the highest, suppress IoU > thr
    (torchvision semantics).  Returns kept indices (in sorted order), at most ``max_keep``."""
    n = boxes.shape[0]
    alive = np.ones(n, bool)
    keep: List[int] = []
    thr = np.float32(iou_thr)
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        if len(keep) >= max_keep:
            break
        if i + 1 < n:
            iou = box_iou_f32(boxes[i], boxes[i + 1:])
            alive[i + 1:] &= ~(iou > thr)
    return np.asarray(keep, np.int64)


def non_max_suppression(pred: np.ndarray, nc: int, conf_thres: float = 0.25, iou_thres: float = 0.7,
                        max_det: int = 300, agnostic: bool = False, multi_label: bool = False) -> List[np.ndarray]:
    """A11.  pred: (B, 4+nc+nm, A) float32.  Returns per image (n, 6+nm):
    [x1,y1,x2,y2,conf,cls, coefs...] sorted by confidence descending (ties: lower anchor index
    first -- a documented choice; upstream's ordering among exact ties is unspecified).
    The upstream wall-clock ``time_limit`` break is deliberately not reproduced (non-deterministic).
    """
    out = []
    pred = pred.astype(np.float32)
    for xi in range(pred.shape[0]):
        x = pred[xi].T  # (A, 4+nc+nm)
        scores = x[:, 4:4 + nc]
        if multi_label and nc > 1:
            # upstream's validator form (multi_label=True): EVERY (anchor, class) pair above the threshold is a candidate,
            # in row-major (anchor, class) order; then the same sort / max_nms cap / class-offset NMS / max_det cut
            ai, ci = np.nonzero(scores > np.float32(conf_thres))
            if ai.size == 0:
                out.append(np.zeros((0, 6 + x.shape[1] - 4 - nc), np.float32))
                continue
            pc = scores[ai, ci]
            o = np.argsort(-pc, kind="stable")[:MAX_NMS]
            ai, ci, pc = ai[o], ci[o], pc[o]
            cx, cy, w, h = x[ai, 0], x[ai, 1], x[ai, 2], x[ai, 3]
            hw, hh = w / np.float32(2), h / np.float32(2)
            xyxy = np.stack((cx - hw, cy - hh, cx + hw, cy + hh), 1).astype(np.float32)
            off = (ci.astype(np.float32) * np.float32(0.0 if agnostic else MAX_WH))[:, None]
            keep = nms_greedy((xyxy + off).astype(np.float32), iou_thres, max_det)
            out.append(np.concatenate((xyxy[keep], pc[keep, None], ci[keep, None].astype(np.float32), x[ai[keep], 4 + nc:]), 1).astype(np.float32))
            continue
        conf = scores.max(1)
        cls = scores.argmax(1)  # first max on ties
        cand = np.nonzero(conf > np.float32(conf_thres))[0]
        if cand.size == 0:
            out.append(np.zeros((0, 6 + x.shape[1] - 4 - nc), np.float32))
            continue
        order = cand[np.argsort(-conf[cand], kind="stable")][:MAX_NMS]
        cx, cy, w, h = x[order, 0], x[order, 1], x[order, 2], x[order, 3]
        hw, hh = w / np.float32(2), h / np.float32(2)
        xyxy = np.stack((cx - hw, cy - hh, cx + hw, cy + hh), 1).astype(np.float32)
        off = (cls[order].astype(np.float32) * np.float32(0.0 if agnostic else MAX_WH))[:, None]
        keep = nms_greedy((xyxy + off).astype(np.float32), iou_thres, max_det)
        sel = order[keep]
        det = np.concatenate((xyxy[keep], conf[sel, None], cls[sel, None].astype(np.float32),
                              x[sel, 4 + nc:]), 1).astype(np.float32)
        out.append(det)
    return out


# ----------------------------------------------------------------------------------------------
# A12: process_mask / crop_mask / scale_boxes
# ----------------------------------------------------------------------------------------------
def process_mask(protos: torch.Tensor, coefs: torch.Tensor, boxes: torch.Tensor,
                 shape: Tuple[int, int]) -> torch.Tensor:
    """A.3 step 2.  protos (nm, mh, mw); coefs (n, nm); boxes (n,4) xyxy in letterboxed pixels;
    shape = (H, W) of the network input.  Returns bool (n, H, W)."""
    c, mh, mw = protos.shape
    ih, iw = shape
    if coefs.shape[0] == 0:
        return torch.zeros((0, ih, iw), dtype=torch.bool)
    m = (coefs.float() @ protos.float().view(c, -1)).view(-1, mh, mw)
    wr, hr = mw / iw, mh / ih
    b = boxes.float().clone()
    b[:, 0] *= wr
    b[:, 2] *= wr
    b[:, 1] *= hr
    b[:, 3] *= hr
    x1, y1, x2, y2 = b[:, 0, None, None], b[:, 1, None, None], b[:, 2, None, None], b[:, 3, None, None]
    r = torch.arange(mw, dtype=torch.float32)[None, None, :]
    cc = torch.arange(mh, dtype=torch.float32)[None, :, None]
    m = m * ((r >= x1) * (r < x2) * (cc >= y1) * (cc < y2))
    m = F.interpolate(m[None], (ih, iw), mode="bilinear", align_corners=False)[0]
    return m > 0.0


def scale_boxes(img1_shape: Tuple[int, int], boxes: np.ndarray, img0_shape: Tuple[int, int]) -> np.ndarray:
    """A.3 step 3: undo letterbox, clip to the original image."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    padx = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pady = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    b = boxes.astype(np.float32).copy()
    b[:, [0, 2]] -= np.float32(padx)
    b[:, [1, 3]] -= np.float32(pady)
    b[:, :4] /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, img0_shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, img0_shape[0])
    return b


def predict(model: SegmentationModel, imgs_u8_hwc: Sequence[np.ndarray], imgsz: int = 640,
            conf: float = 0.25, iou: float = 0.7, max_det: int = 300):
    """A1: end-to-end CPU predict.  Returns list of dict(boxes (n,6) in original px, masks bool
    (n,imgsz,imgsz), det_letterboxed (n,6+32))."""
    lb = [letterbox(im, (imgsz, imgsz))[0] for im in imgs_u8_hwc]
    x = preprocess(lb)
    model.eval()
    with torch.no_grad():
        preds, protos = model(x)
    dets = non_max_suppression(preds.numpy(), model.nc, conf, iou, max_det)
    res = []
    for i, d in enumerate(dets):
        masks = process_mask(protos[i], torch.from_numpy(d[:, 6:]), torch.from_numpy(d[:, :4]),
                             (x.shape[2], x.shape[3]))
        boxes = d[:, :6].copy()
        boxes[:, :4] = scale_boxes((x.shape[2], x.shape[3]), d[:, :4], imgs_u8_hwc[i].shape[:2])
        res.append({"boxes": boxes, "masks": masks.numpy(), "det_letterboxed": d})
    return res
