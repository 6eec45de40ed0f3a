"""``Results`` / ``Boxes`` / ``Masks`` containers with the API surface the reference scripts consume.

Call sites honoured: ``print(results)`` (/root/reference/BscanBased/yolo8_seg_predict.py:9);
``for box in res.boxes: box.xyxy.tolist()[0]; float(box.conf); int(box.cls)``
(BscanBased/yolo/yolo_eval.py:30-35, yolo/yolo_folder_eval.py:18-24); ``res.names = {...}`` assignable
(yolo_folder_eval.py:26); ``res.plot()`` -> BGR uint8 ndarray usable by cv2.imshow / cv2.imwrite
(yolo_eval.py:37-38); ``box.xyxy[0].cpu().numpy()`` and ``model.names[int(box.cls[0])]``
(signals/improved_multisignal/visualization/yolo_detector.py:48-51).
Results own CPU tensors.  Drawing uses PIL / numpy (cv2 is not available here).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

_PALETTE = [(56, 56, 255), (151, 157, 255), (31, 112, 255), (29, 178, 255), (49, 210, 207), (10, 249, 72),
            (23, 204, 146), (134, 219, 61), (52, 147, 26), (187, 212, 0), (168, 153, 44), (255, 194, 0)]  # BGR


class _TensorView:
    """Shared helpers of Boxes / Masks (``.cpu()``, ``.numpy()``, ``.to()``, ``len``, ``shape``)."""

    def __init__(self, data: torch.Tensor, orig_shape: Tuple[int, int]):
        self.data = data
        self.orig_shape = tuple(orig_shape)

    @property
    def shape(self):
        return self.data.shape

    def __len__(self):
        return self.data.shape[0]

    def _new(self, data):
        return self.__class__(data, self.orig_shape)

    def cpu(self):
        return self._new(self.data.cpu())

    def numpy(self):
        return self._new(self.data.cpu()).data.numpy()

    def cuda(self):
        return self._new(self.data.cuda())

    def to(self, *a, **k):
        return self._new(self.data.to(*a, **k))

    def __getitem__(self, idx):
        d = self.data[idx]
        if d.dim() == self.data.dim() - 1:
            d = d[None]
        return self._new(d)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class Boxes(_TensorView):
    """``data`` (n,6) float32: x1,y1,x2,y2 (original-image pixels), confidence, class."""

    @property
    def xyxy(self) -> torch.Tensor:
        return self.data[:, :4]

    @property
    def conf(self) -> torch.Tensor:
        return self.data[:, 4]

    @property
    def cls(self) -> torch.Tensor:
        return self.data[:, 5]

    @property
    def xywh(self) -> torch.Tensor:
        b = self.xyxy
        return torch.stack(((b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), 1)

    @property
    def xyxyn(self) -> torch.Tensor:
        h, w = self.orig_shape
        return self.xyxy / torch.tensor([w, h, w, h], dtype=self.data.dtype, device=self.data.device)

    @property
    def xywhn(self) -> torch.Tensor:
        h, w = self.orig_shape
        return self.xywh / torch.tensor([w, h, w, h], dtype=self.data.dtype, device=self.data.device)

    def __repr__(self):
        return (f"Boxes(n={len(self)}, orig_shape={self.orig_shape})\ncls: {self.cls}\nconf: {self.conf}\n"
                f"xyxy: {self.xyxy}")


def _trace_polygon(mask: np.ndarray) -> np.ndarray:
    """Outer boundary of the largest 4-connected foreground run structure, as an (k,2) float32 x,y polygon.
    Minimal Moore-neighbour tracing (cv2.findContours is unavailable)."""
    m = np.pad(mask.astype(bool), 1)
    ys, xs = np.nonzero(m)
    if ys.size == 0:
        return np.zeros((0, 2), np.float32)
    start = (int(ys[0]), int(xs[np.nonzero(ys == ys[0])[0][0]]))
    nbrs = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]  # clockwise from east
    pts = [start]
    cur, prev_dir = start, 6
    for _ in range(4 * (m.shape[0] + m.shape[1]) + ys.size):
        found = False
        for k in range(8):
            d = (prev_dir + 1 + k) % 8
            ny, nx = cur[0] + nbrs[d][0], cur[1] + nbrs[d][1]
            if m[ny, nx]:
                cur = (ny, nx)
                prev_dir = (d + 4) % 8
                found = True
                break
        if not found or cur == start:
            break
        pts.append(cur)
    p = np.asarray(pts, np.float32)
    return np.stack((p[:, 1] - 1, p[:, 0] - 1), 1)


class Masks(_TensorView):
    """``data`` (n,H,W) uint8 {0,1} at network-input (letterboxed) resolution, like upstream."""

    def __init__(self, data: torch.Tensor, orig_shape, pad=(0, 0), gain: float = 1.0):
        super().__init__(data, orig_shape)
        self._pad, self._gain = pad, gain

    def _new(self, data):
        return Masks(data, self.orig_shape, self._pad, self._gain)

    @property
    def xy(self) -> List[np.ndarray]:
        """Polygons in original-image pixel coordinates."""
        out = []
        for m in self.data.cpu().numpy():
            p = _trace_polygon(m)
            if p.size:
                p = (p - np.asarray(self._pad, np.float32)) / np.float32(self._gain)
                p[:, 0] = p[:, 0].clip(0, self.orig_shape[1])
                p[:, 1] = p[:, 1].clip(0, self.orig_shape[0])
            out.append(p)
        return out

    @property
    def xyn(self) -> List[np.ndarray]:
        h, w = self.orig_shape
        return [p / np.asarray([w, h], np.float32) if p.size else p for p in self.xy]

    def __repr__(self):
        return f"Masks(n={len(self)}, shape={tuple(self.data.shape)}, orig_shape={self.orig_shape})"


class Results:
    def __init__(self, orig_img: np.ndarray, path: str, names: Dict[int, str], boxes: torch.Tensor,
                 masks: Optional[torch.Tensor] = None, speed: Optional[Dict[str, float]] = None,
                 net_shape: Optional[Tuple[int, int]] = None):
        self.orig_img = orig_img
        self.orig_shape = orig_img.shape[:2]
        self.path = path
        self.names = dict(names)
        self.boxes = Boxes(boxes, self.orig_shape)
        net_shape = tuple(net_shape) if net_shape else self.orig_shape
        gain = min(net_shape[0] / self.orig_shape[0], net_shape[1] / self.orig_shape[1])
        pad = (round((net_shape[1] - self.orig_shape[1] * gain) / 2 - 0.1),
               round((net_shape[0] - self.orig_shape[0] * gain) / 2 - 0.1))
        self.masks = Masks(masks, self.orig_shape, pad, gain) if masks is not None else None
        self.probs = None
        self.keypoints = None
        self.obb = None
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}
        self.save_dir = None
        self._net_shape = net_shape
        self._pad, self._gain = pad, gain

    def __len__(self):
        return len(self.boxes)

    def cpu(self):
        return self

    def verbose(self) -> str:
        if len(self) == 0:
            return "(no detections), "
        cls = self.boxes.cls.to(torch.int64)
        parts = []
        for c in cls.unique().tolist():
            n = int((cls == c).sum())
            parts.append(f"{n} {self.names.get(int(c), str(c))}{'s' * (n > 1)}")
        return ", ".join(parts) + ", "

    def __repr__(self):
        return (f"mi355yolo.Results object with attributes:\n\nboxes: {type(self.boxes).__module__}.Boxes object "
                f"({len(self.boxes)} boxes)\nmasks: "
                f"{'None' if self.masks is None else f'Masks object {tuple(self.masks.data.shape)}'}\n"
                f"names: {self.names}\norig_shape: {self.orig_shape}\npath: '{self.path}'\n"
                f"save_dir: {self.save_dir!r}\nspeed: {self.speed}\nsummary: {self.verbose()}")

    __str__ = __repr__

    def summary(self) -> List[dict]:
        out = []
        for i in range(len(self.boxes)):
            b = self.boxes.data[i].tolist()
            out.append({"name": self.names.get(int(b[5]), str(int(b[5]))), "class": int(b[5]), "confidence": b[4],
                        "box": {"x1": b[0], "y1": b[1], "x2": b[2], "y2": b[3]}})
        return out

    # ------------------------------------------------------------------ drawing
    def _masks_on_original(self) -> Optional[np.ndarray]:
        """(n, h0, w0) bool masks cropped out of the letterboxed frame and nearest-resized to the original."""
        if self.masks is None or len(self.masks) == 0:
            return None
        m = self.masks.data.cpu().numpy().astype(bool)
        h0, w0 = self.orig_shape
        px, py = self._pad
        uh, uw = int(round(h0 * self._gain)), int(round(w0 * self._gain))
        m = m[:, py:py + uh, px:px + uw]
        yi = np.clip(((np.arange(h0) + 0.5) * (m.shape[1] / h0)).astype(np.int64), 0, m.shape[1] - 1)
        xi = np.clip(((np.arange(w0) + 0.5) * (m.shape[2] / w0)).astype(np.int64), 0, m.shape[2] - 1)
        return m[:, yi][:, :, xi]

    def plot(self, conf: bool = True, labels: bool = True, boxes: bool = True, masks: bool = True,
             line_width: Optional[int] = None) -> np.ndarray:
        """Annotated copy of the original image, BGR uint8 (upstream's return convention)."""
        from PIL import Image, ImageDraw
        img = np.ascontiguousarray(self.orig_img.copy())
        lw = line_width or max(round(sum(img.shape[:2]) / 2 * 0.003), 2)
        if masks:
            mm = self._masks_on_original()
            if mm is not None:
                f = img.astype(np.float32)
                for i, m in enumerate(mm):
                    col = np.asarray(_PALETTE[int(self.boxes.cls[i]) % len(_PALETTE)], np.float32)
                    f[m] = f[m] * 0.5 + col * 0.5
                img = f.clip(0, 255).astype(np.uint8)
        if boxes and len(self.boxes):
            pil = Image.fromarray(img[:, :, ::-1].copy())
            dr = ImageDraw.Draw(pil)
            for i in range(len(self.boxes)):
                x1, y1, x2, y2, cf, c = self.boxes.data[i].tolist()
                col = tuple(reversed(_PALETTE[int(c) % len(_PALETTE)]))
                dr.rectangle([x1, y1, x2, y2], outline=col, width=lw)
                if labels:
                    text = self.names.get(int(c), str(int(c))) + (f" {cf:.2f}" if conf else "")
                    tw = dr.textlength(text)
                    ty = y1 - 12 if y1 >= 12 else y1 + 1
                    dr.rectangle([x1, ty, x1 + tw + 4, ty + 12], fill=col)
                    dr.text((x1 + 2, ty), text, fill=(255, 255, 255))
            img = np.ascontiguousarray(np.asarray(pil)[:, :, ::-1])
        return img

    def save(self, filename: Optional[str] = None) -> str:
        from PIL import Image
        filename = filename or f"results_{os.path.basename(self.path)}"
        os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
        Image.fromarray(self.plot()[:, :, ::-1]).save(filename)
        return filename
