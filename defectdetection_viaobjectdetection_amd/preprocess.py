"""Host-side source loading + LetterBox pre-processing.

Stands where upstream's ``LoadImagesAndVideos`` + ``LetterBox`` stand in
``SegmentationPredictor.preprocess`` (call site /root/reference/BscanBased/yolo8_seg_predict.py:8).
``cv2`` is not available: PNG/JPEG decode goes through PIL and the bilinear resize is a numpy
restatement of ``cv2.resize(..., INTER_LINEAR)`` on uint8 (half-pixel centres, no antialias, rounded to
nearest).  The /255 normalisation and HWC->planar step are NOT done here: the engine consumes uint8
NHWC and folds 1/255 into the stem convolution.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple, Union

import numpy as np

IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")


def load_image(path: str) -> np.ndarray:
    """Decode to HxWx3 uint8 in BGR order (what ``cv2.imread`` returns; gray PNGs replicate the channel)."""
    from PIL import Image
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} does not exist")
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def to_bgr_array(src) -> np.ndarray:
    """Accept a path, a PIL image (RGB) or an HxW / HxWx3 uint8 ndarray (taken as BGR like upstream)."""
    if isinstance(src, (str, os.PathLike)):
        return load_image(os.fspath(src))
    if isinstance(src, np.ndarray):
        a = src
        if a.dtype != np.uint8:
            raise TypeError("ndarray sources must be uint8")
        if a.ndim == 2:
            a = np.repeat(a[:, :, None], 3, axis=2)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected HxW or HxWx3, got {a.shape}")
        return np.ascontiguousarray(a)
    try:
        from PIL import Image
        if isinstance(src, Image.Image):
            return np.ascontiguousarray(np.asarray(src.convert("RGB"), dtype=np.uint8)[:, :, ::-1])
    except ImportError:
        pass
    raise TypeError(f"unsupported source type {type(src)}")


def expand_sources(source) -> Tuple[List[np.ndarray], List[str]]:
    """source: path | directory | list of those / arrays.  Returns (BGR images, display paths)."""
    items: List = []
    if isinstance(source, (list, tuple)):
        items = list(source)
    elif isinstance(source, (str, os.PathLike)) and os.path.isdir(source):
        items = sorted(os.path.join(source, f) for f in os.listdir(source) if f.lower().endswith(IMG_EXT))
        if not items:
            raise FileNotFoundError(f"no images found in {source}")
    else:
        items = [source]
    imgs, paths = [], []
    for i, it in enumerate(items):
        imgs.append(to_bgr_array(it))
        paths.append(os.fspath(it) if isinstance(it, (str, os.PathLike)) else f"image{i}.jpg")
    return imgs, paths


def resize_linear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Bilinear resize of uint8 HxWxC with half-pixel centres and edge replication."""
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()
    sy, sx = h / new_h, w / new_w
    yy = (np.arange(new_h) + 0.5) * sy - 0.5
    xx = (np.arange(new_w) + 0.5) * sx - 0.5
    y0 = np.floor(yy).astype(np.int64)
    x0 = np.floor(xx).astype(np.int64)
    wy = (yy - y0)[:, None, None]  # float64: the exact bilinear value, rounded to nearest once
    wx = (xx - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    f = img.astype(np.float64)
    r0 = f[y0c]
    r1 = f[y1c]
    top = r0[:, x0c] * (1 - wx) + r0[:, x1c] * wx
    bot = r1[:, x0c] * (1 - wx) + r1[:, x1c] * wx
    out = top * (1 - wy) + bot * wy
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def letterbox_shape(shape: Tuple[int, int], imgsz: Tuple[int, int], auto: bool, stride: int = 32):
    """Geometry of A.2: returns (ratio, (unpad_h, unpad_w), (top, bottom, left, right), (out_h, out_w))."""
    h, w = shape
    r = min(imgsz[0] / h, imgsz[1] / w)
    unpad_w, unpad_h = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz[1] - unpad_w, imgsz[0] - unpad_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, (unpad_h, unpad_w), (top, bottom, left, right), (unpad_h + top + bottom, unpad_w + left + right)


def letterbox(img: np.ndarray, imgsz: Tuple[int, int] = (640, 640), auto: bool = False, stride: int = 32,
              color: int = 114) -> np.ndarray:
    _, (uh, uw), (top, bottom, left, right), (oh, ow) = letterbox_shape(img.shape[:2], imgsz, auto, stride)
    if img.shape[:2] != (uh, uw):
        img = resize_linear_u8(img, uh, uw)
    out = np.full((oh, ow, 3), color, np.uint8)
    out[top:top + uh, left:left + uw] = img
    return out


def scale_boxes_to_original(boxes_xyxy: np.ndarray, net_shape: Tuple[int, int], orig_shape: Tuple[int, int]) -> np.ndarray:
    """A.3 step 3: letterboxed-pixel boxes -> original-image pixels, clipped."""
    gain = min(net_shape[0] / orig_shape[0], net_shape[1] / orig_shape[1])
    padx = round((net_shape[1] - orig_shape[1] * gain) / 2 - 0.1)
    pady = round((net_shape[0] - orig_shape[0] * gain) / 2 - 0.1)
    b = boxes_xyxy.astype(np.float32).copy()
    b[:, [0, 2]] -= np.float32(padx)
    b[:, [1, 3]] -= np.float32(pady)
    b /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, orig_shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, orig_shape[0])
    return b
