"""Polygon-label segmentation dataset for ``YOLO.train`` / ``YOLO.val`` (SURVEY.md A2, Appendix A.4 "labels").

Stands where upstream's ``YOLODataset(task='segment')`` + ``build_dataloader`` stand (reached from
/root/reference/BscanBased/yolo_seg_train.py:12 with ``data-seg.yaml``:1-5).  Layout contract is upstream's:
``<root>/images/<split>/x.png`` pairs with ``<root>/labels/<split>/x.txt`` whose rows are
``cls x1 y1 x2 y2 ...`` (normalised polygon) or ``cls cx cy w h`` (box; becomes a 4-point polygon).

B-scan datasets are small (hundreds of 320x320 PNGs), so the whole split is decoded and letterboxed once and kept
in host memory as uint8; a batch is a gather + one H2D copy.  Augmentation here is the left-right flip only --
mosaic / random affine / HSV (Appendix A.4) are next-row N2 (GPU-side augmentation) and not built.

``write_polygon_dataset`` is the fixed converter the survey calls for (D5): the reference's
``yolo_ds_segmentation.py``:79-96 writes PNG masks the trainer cannot read and slices ``x_min:x_max`` although most
annotation boxes have ``x_min > x_max``; this one sorts the corners and writes polygon ``.txt`` rows.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .preprocess import IMG_EXT, letterbox, letterbox_shape, load_image


def read_data_yaml(path: str) -> Dict:
    import yaml
    if not os.path.isfile(path):
        raise FileNotFoundError(f"dataset yaml '{path}' does not exist")
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    root = cfg.get("path") or os.path.dirname(os.path.abspath(path))
    if not os.path.isabs(root):
        root = os.path.join(os.path.dirname(os.path.abspath(path)), root)
    names = cfg.get("names")
    if isinstance(names, (list, tuple)):
        names = {i: n for i, n in enumerate(names)}
    if not names:
        nc = int(cfg.get("nc", 0))
        if nc <= 0:
            raise ValueError(f"{path}: neither 'names' nor 'nc' given")
        names = {i: f"class{i}" for i in range(nc)}
    names = {int(k): str(v) for k, v in names.items()}
    out = {"root": root, "names": names, "nc": len(names)}
    for split in ("train", "val", "test"):
        v = cfg.get(split)
        if v:
            out[split] = v if os.path.isabs(v) else os.path.join(root, v)
    if "train" not in out:
        raise ValueError(f"{path}: 'train' split is missing")
    out.setdefault("val", out["train"])
    return out


def img2label_path(img_path: str) -> str:
    sa, sb = f"{os.sep}images{os.sep}", f"{os.sep}labels{os.sep}"
    i = img_path.rfind(sa)
    stem = img_path if i < 0 else img_path[:i] + sb + img_path[i + len(sa):]
    return os.path.splitext(stem)[0] + ".txt"


def parse_label_file(path: str) -> List[Tuple[int, np.ndarray]]:
    """Rows -> (cls, (n,2) normalised polygon).  A missing file is a background image (no instances)."""
    out: List[Tuple[int, np.ndarray]] = []
    if not os.path.isfile(path):
        return out
    with open(path) as f:
        for ln, row in enumerate(f, 1):
            v = row.split()
            if not v:
                continue
            nums = [float(t) for t in v[1:]]
            c = int(float(v[0]))
            if len(nums) == 4:                                   # box row
                cx, cy, w, h = nums
                poly = np.array([[cx - w / 2, cy - h / 2], [cx + w / 2, cy - h / 2], [cx + w / 2, cy + h / 2],
                                 [cx - w / 2, cy + h / 2]], np.float64)
            elif len(nums) >= 6 and len(nums) % 2 == 0:
                poly = np.array(nums, np.float64).reshape(-1, 2)
            else:
                raise ValueError(f"{path}:{ln}: expected 'cls cx cy w h' or 'cls x1 y1 x2 y2 x3 y3 ...', got {len(nums)} numbers")
            if c < 0 or poly.min() < -1e-6 or poly.max() > 1 + 1e-6:
                raise ValueError(f"{path}:{ln}: class must be >= 0 and coordinates normalised to [0, 1]")
            out.append((c, np.clip(poly, 0.0, 1.0)))
    return out


def rasterize_polygon(poly_px: np.ndarray, h: int, w: int) -> np.ndarray:
    """Even-odd scanline fill sampled at pixel centres; returns (h,w) bool."""
    from PIL import Image, ImageDraw
    im = Image.new("L", (w, h), 0)
    ImageDraw.Draw(im).polygon([(float(x), float(y)) for x, y in poly_px], outline=1, fill=1)
    return np.asarray(im, dtype=np.uint8) > 0


def overlap_mask(polys_px: Sequence[np.ndarray], imgsz: Tuple[int, int], ratio: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """A.4 "GT masks": one (H/ratio, W/ratio) map whose value is (instance index + 1) with instances ordered by mask area
    descending, later (smaller) instances overwriting earlier ones.  uint8, or int32 when an image holds more than 255
    instances (upstream's rule; a uint8 map would wrap around).  Returns (map, order)."""
    H, W = imgsz
    mh, mw = H // ratio, W // ratio
    small = []
    for p in polys_px:
        full = rasterize_polygon(p, H, W).astype(np.float32)
        small.append(full.reshape(mh, ratio, mw, ratio).mean((1, 3)) >= 0.5)
    order = np.argsort([-int(m.sum()) for m in small], kind="stable") if small else np.zeros(0, np.int64)
    out = np.zeros((mh, mw), np.uint8 if len(small) <= 255 else np.int32)
    for rank, j in enumerate(order):
        out[small[j]] = rank + 1
    return out, order


class SegDataset:
    def __init__(self, img_dir: str, imgsz: int = 640, nc: Optional[int] = None):
        if not os.path.isdir(img_dir):
            raise FileNotFoundError(f"image directory '{img_dir}' does not exist")
        self.files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(img_dir) for f in fs if f.lower().endswith(IMG_EXT))
        if not self.files:
            raise FileNotFoundError(f"no images under '{img_dir}'")
        self.imgsz = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
        H, W = self.imgsz
        self.images = np.empty((len(self.files), H, W, 3), np.uint8)            # RGB, letterboxed
        self.orig_shapes: List[Tuple[int, int]] = []
        self.labels: List[List[Tuple[int, np.ndarray]]] = []                     # polygons in letterboxed pixels
        for i, f in enumerate(self.files):
            bgr = load_image(f)
            h0, w0 = bgr.shape[:2]
            r, (uh, uw), (top, _, left, _), _ = letterbox_shape((h0, w0), self.imgsz, auto=False)
            self.images[i] = letterbox(bgr, self.imgsz, auto=False)[:, :, ::-1]
            self.orig_shapes.append((h0, w0))
            inst = []
            for c, poly in parse_label_file(img2label_path(f)):
                if nc is not None and c >= nc:
                    raise ValueError(f"{img2label_path(f)}: class {c} >= nc {nc}")
                inst.append((c, poly * np.array([uw, uh], np.float64) + np.array([left, top], np.float64)))
            self.labels.append(inst)

    def __len__(self) -> int:
        return len(self.files)

    def batch(self, indices: Sequence[int], flip: Optional[Sequence[bool]] = None) -> Dict[str, np.ndarray]:
        """Collate: img (B,H,W,3) uint8 RGB; batch_idx (N,), cls (N,), bboxes (N,4) xywh normalised to the network
        input (box = polygon bounds), masks (B,H/4,W/4) overlap-encoded, instances of an image sorted like the map."""
        H, W = self.imgsz
        imgs = self.images[list(indices)]
        bidx, cls, boxes = [], [], []
        masks = np.zeros((len(indices), H // 4, W // 4), np.uint8)
        if any(len(self.labels[i]) > 255 for i in indices):
            masks = masks.astype(np.int32)
        for b, i in enumerate(indices):
            polys = [p for _, p in self.labels[i]]
            if flip is not None and flip[b]:
                imgs[b] = imgs[b][:, ::-1]
                polys = [np.stack((W - p[:, 0], p[:, 1]), 1) for p in polys]
            if not polys:
                continue
            masks[b], order = overlap_mask(polys, self.imgsz)
            for j in order:
                p = polys[j]
                x1, y1, x2, y2 = p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()
                bidx.append(b)
                cls.append(self.labels[i][j][0])
                boxes.append([(x1 + x2) / 2 / W, (y1 + y2) / 2 / H, (x2 - x1) / W, (y2 - y1) / H])
        return {"img": np.ascontiguousarray(imgs), "batch_idx": np.asarray(bidx, np.float32),
                "cls": np.asarray(cls, np.float32), "bboxes": np.asarray(boxes, np.float32).reshape(-1, 4), "masks": masks}


def epoch_batches(n: int, batch: int, epoch: int, seed: int = 0, rank: int = 0, world: int = 1, shuffle: bool = True):
    """Index lists of one epoch.  Every rank draws the same permutation and takes its contiguous share of each
    global batch (global batch = batch * world); the tail is wrapped so that all ranks run the same number of steps."""
    rng = np.random.default_rng(seed + epoch)
    perm = rng.permutation(n) if shuffle else np.arange(n)
    gb = batch * world
    steps = max(1, math.ceil(n / gb))
    perm = np.resize(perm, steps * gb)
    return [perm[s * gb + rank * batch: s * gb + (rank + 1) * batch].tolist() for s in range(steps)]


def write_polygon_dataset(annotations: Dict, image_root: str, out_root: str, class_names: Optional[Sequence[str]] = None,
                          val_fraction: float = 0.2, seed: int = 0) -> str:
    """``annotations.json`` semantics (``{folder: {file: [{"bbox": [x_min, x_max, y_min, y_max], "label": str}]}}``,
    x often reversed) -> ``out_root/{images,labels}/{train,val}`` + ``data-seg.yaml``; returns the yaml path.
    Images are linked (or copied when linking fails) from ``image_root/<folder>/<file>``."""
    import shutil
    import yaml
    from PIL import Image
    labels = sorted({a["label"] for fo in annotations.values() for items in fo.values() for a in items})
    names = list(class_names) if class_names else labels
    items = [(fo, fi) for fo in sorted(annotations) for fi in sorted(annotations[fo])]
    rng = np.random.default_rng(seed)
    is_val = np.zeros(len(items), bool)
    is_val[rng.permutation(len(items))[:int(round(len(items) * val_fraction))]] = True
    for split in ("train", "val"):
        os.makedirs(os.path.join(out_root, "images", split), exist_ok=True)
        os.makedirs(os.path.join(out_root, "labels", split), exist_ok=True)
    for (fo, fi), v in zip(items, is_val):
        src = os.path.join(image_root, fo, fi)
        if not os.path.isfile(src):
            raise FileNotFoundError(src)
        with Image.open(src) as im:
            w, h = im.size
        split = "val" if v else "train"
        stem = f"{fo}_{os.path.splitext(fi)[0]}"
        dst = os.path.join(out_root, "images", split, stem + os.path.splitext(fi)[1])
        if not os.path.exists(dst):
            try:
                os.symlink(os.path.abspath(src), dst)
            except OSError:
                shutil.copyfile(src, dst)
        rows = []
        for a in annotations[fo][fi]:
            xa, xb, ya, yb = a["bbox"]
            x1, x2 = sorted((min(max(xa, 0), w), min(max(xb, 0), w)))
            y1, y2 = sorted((min(max(ya, 0), h), min(max(yb, 0), h)))
            if x2 - x1 < 1 or y2 - y1 < 1:
                continue
            c = names.index(a["label"]) if a["label"] in names else 0
            pts = [(x1, y1), (x2, y1), (x2, y2), (x1, y2)]
            rows.append(f"{c} " + " ".join(f"{x / w:.6f} {y / h:.6f}" for x, y in pts))
        with open(os.path.join(out_root, "labels", split, stem + ".txt"), "w") as f:
            f.write("\n".join(rows) + ("\n" if rows else ""))
    ypath = os.path.join(out_root, "data-seg.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"path": os.path.abspath(out_root), "train": "images/train", "val": "images/val",
                        "names": {i: n for i, n in enumerate(names)}}, f, sort_keys=False)
    return ypath
