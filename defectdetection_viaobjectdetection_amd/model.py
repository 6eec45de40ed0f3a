"""``YOLO`` -- the model facade the reference scripts call, backed by the libmi355yolo engine.

Mirrors the surface of ``ultralytics.YOLO`` as exercised by the reference:
  ``YOLO("....yaml")`` / ``YOLO("....pt")``            /root/reference/BscanBased/yolo_seg_train.py:7-8,
                                                       yolo8_seg_predict.py:5
  ``model.predict(path, save=True)`` / ``model(path, conf=..)``   yolo8_seg_predict.py:8, yolo_detector.py:40
  ``model.names``                                      yolo_detector.py:51
  ``model.train(data=, epochs=, imgsz=, project=, name=, device=)``   yolo_seg_train.py:12-19
Nothing is ever downloaded: a model given by NAME that is not a local file raises a clear offline error.
All network arithmetic runs in the HIP kernels behind the C-ABI; there is no CPU fallback.
"""
from __future__ import annotations

import math
import os
import re
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .preprocess import expand_sources, letterbox, letterbox_shape, scale_boxes_to_original
from .results import Results
from .spec import SCALES, V9C, conv_specs, count_parameters, init_state_dict, state_dict_keys

_YAML_RE = re.compile(r"^yolov8([nsmlx])?-seg\.ya?ml$")
_V9C_RE = re.compile(r"^yolov9c-seg\.ya?ml$")
CKPT_FORMAT = "mi355yolo-seg-v1"


class OfflineModelError(FileNotFoundError):
    pass


def _increment_dir(base: str) -> str:
    if not os.path.exists(base):
        return base
    i = 2
    while os.path.exists(f"{base}{i}"):
        i += 1
    return f"{base}{i}"


class YOLO:
    def __init__(self, model: str = "yolov8s-seg.yaml", task: Optional[str] = None, verbose: bool = False):
        self.task = "segment"
        self.ckpt_path: Optional[str] = None
        self.overrides: Dict = {}
        self._engines: Dict[Tuple[int, int, int], object] = {}
        self.train_args: Dict = {}
        name = os.path.basename(str(model))
        m = _YAML_RE.match(name)
        if m:
            self.scale = m.group(1) or "n"
            self.nc = 80
            self.names = {i: f"class{i}" for i in range(self.nc)}
            if os.path.isfile(model):
                self._read_yaml_overrides(model)
            self.state_dict = init_state_dict(self.scale, self.nc, seed=0)
        elif _V9C_RE.match(name):
            # the architecture the reference scripts literally name (/root/reference/BscanBased/yolo_seg_train.py:7)
            self.scale = V9C
            self.nc = 80
            self.names = {i: f"class{i}" for i in range(self.nc)}
            if os.path.isfile(model):
                self._read_yaml_overrides(model)
            self.state_dict = init_state_dict(self.scale, self.nc, seed=0)
        elif name.endswith((".yaml", ".yml")):
            raise NotImplementedError(
                f"architecture '{name}' is not built: this package implements the YOLOv8{{n,s,m,l,x}}-seg and YOLOv9c-seg "
                "graphs; the detect-task families (yolov5u / yolo11) are listed as next rows in SURVEY.md 8(f) N4")
        elif name.endswith(".pt"):
            if not os.path.isfile(model):
                raise OfflineModelError(
                    f"'{model}' is not a local file.  Upstream would download weights by name; this build is offline "
                    "and never fetches.  Pass a path to a checkpoint written by this package (YOLO.save / .train).")
            self._load_checkpoint(model)
        else:
            raise ValueError(f"unsupported model specifier '{model}' (expected *.yaml or *.pt)")

    # ------------------------------------------------------------------ model state
    def _read_yaml_overrides(self, path: str) -> None:
        import yaml
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
        if "nc" in cfg:
            self.set_classes(int(cfg["nc"]))

    def set_classes(self, nc: int, names: Optional[Dict[int, str]] = None) -> None:
        """Re-initialise the head for ``nc`` classes (what upstream does when data.yaml disagrees with the yaml)."""
        self.nc = nc
        self.names = dict(names) if names else {i: f"class{i}" for i in range(nc)}
        self.state_dict = init_state_dict(self.scale, nc, seed=0)
        self._drop_engines()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        keys = state_dict_keys(self.scale, self.nc)
        missing = [k for k in keys if k not in sd]
        if missing:
            raise KeyError(f"state dict misses {len(missing)} keys, e.g. {missing[:3]}")
        for s in conv_specs(self.scale, self.nc):
            if s.rep:
                continue
            k = f"{s.name}.conv.weight" if s.has_bn else f"{s.name}.weight"
            if tuple(sd[k].shape) != s.weight_shape:
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {s.weight_shape}")
        self.state_dict = {k: sd[k].detach().cpu().clone() for k in keys}
        self._drop_engines()

    def _load_checkpoint(self, path: str) -> None:
        import pickle
        ck = None
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:  # torch's weights-only loader met a class: not one of ours, maybe upstream's
            ck = None
        if not isinstance(ck, dict) or ck.get("format") != CKPT_FORMAT:
            # a checkpoint written by upstream Ultralytics: recover the tensors without its classes (upstream_ckpt.py)
            from .upstream_ckpt import load_upstream_checkpoint
            up = load_upstream_checkpoint(path)
            self.scale, self.nc, self.names = up["scale"], up["nc"], up["names"]
            self.train_args = up["train_args"]
            self.state_dict = up["state_dict"]
            self.ckpt_path = path
            return
        self.scale, self.nc = ck["scale"], int(ck["nc"])
        self.names = {int(k): v for k, v in ck["names"].items()}
        self.train_args = ck.get("train_args", {})
        self.state_dict = ck["model"]
        self.ckpt_path = path
        self._resume_state = ck.get("trainer")       # present in weights/last.pt: lets train(resume=True) continue

    def save(self, path: str, upstream: bool = False) -> str:
        """``upstream=True`` writes upstream Ultralytics' own checkpoint layout (a pickled ``SegmentationModel`` module
        graph, fp16, upstream_export.py) so that ``ultralytics.YOLO(path)`` of the real package can load it; the default
        is this package's plain state-dict format."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if upstream:
            from .upstream_export import export_upstream_checkpoint
            return export_upstream_checkpoint(path, self.scale, self.nc, self.names, self.state_dict, self.train_args)
        torch.save({"format": CKPT_FORMAT, "scale": self.scale, "nc": self.nc, "names": self.names,
                    "train_args": self.train_args, "model": self.state_dict}, path)
        return path

    def info(self) -> Tuple[int, int]:
        n = count_parameters(self.state_dict)
        return len(conv_specs(self.scale, self.nc)), n

    def _drop_engines(self) -> None:
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def _engine(self, shape: Tuple[int, int], batch: int, device: int):
        from .engine import SegEngine  # loads libmi355yolo.so; raises loudly when it is missing
        key = (shape[0], shape[1], device)
        eng = self._engines.get(key)
        if eng is None or eng.max_batch < batch:
            if eng is not None:
                eng.close()
            eng = SegEngine(self.scale, self.nc, shape, max_batch=max(batch, 1), device=device, keep_raw=False)
            eng.load_state_dict(self.state_dict)
            self._engines[key] = eng
        return eng

    # ------------------------------------------------------------------ inference
    def predict(self, source=None, save: bool = False, imgsz=None, conf: float = 0.25, iou: float = 0.7,
                max_det: int = 300, device=0, verbose: bool = True, retina_masks: bool = False,
                project: Optional[str] = None, name: Optional[str] = None, batch: int = 32, **kwargs) -> List[Results]:
        if source is None:
            raise ValueError("source is required")
        if retina_masks:
            raise NotImplementedError("retina_masks=True is not implemented")
        dev = int(device[0] if isinstance(device, (list, tuple)) else device)
        if imgsz is None:
            imgsz = self.train_args.get("imgsz", 640)  # D7: a checkpoint keeps its training size
        if isinstance(imgsz, int):
            imgsz = (imgsz, imgsz)
        imgsz = tuple(int(math.ceil(s / 32) * 32) for s in imgsz)
        t0 = time.perf_counter()
        imgs, paths = expand_sources(source)
        # auto=True (min-rectangle) letterbox like upstream's predictor for a single shape; mixed shapes pad to imgsz
        shapes = {im.shape[:2] for im in imgs}
        auto = len(shapes) == 1
        net_shapes = [letterbox_shape(im.shape[:2], imgsz, auto)[3] for im in imgs]
        net_shape = net_shapes[0] if auto else imgsz
        lb = [letterbox(im, imgsz, auto=auto) for im in imgs]
        results: List[Results] = []
        save_dir = None
        if save:
            save_dir = _increment_dir(os.path.join(project or os.path.join("runs", "segment"), name or "predict"))
            os.makedirs(save_dir, exist_ok=True)
        t_pre = (time.perf_counter() - t0) * 1e3 / len(imgs)
        for i0 in range(0, len(imgs), batch):
            chunk = lb[i0:i0 + batch]
            eng = self._engine(net_shape, len(chunk), dev)
            with torch.cuda.device(eng.device):
                t1 = time.perf_counter()
                x = torch.from_numpy(np.stack(chunk)[:, :, :, ::-1].copy()).to(eng.device)  # BGR -> RGB, H2D
                preds, protos = eng.forward(x)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                dets, counts, masks = eng.postprocess(preds, protos, conf, iou, max_det, masks=True)
                counts_h = counts.cpu().tolist()
                t3 = time.perf_counter()
            for j, n in enumerate(counts_h):
                d = dets[j, :n, :6].cpu().numpy()
                orig = imgs[i0 + j]
                d[:, :4] = scale_boxes_to_original(d[:, :4], net_shape, orig.shape[:2])
                m = masks[j, :n].cpu()
                speed = {"preprocess": t_pre, "inference": (t2 - t1) * 1e3 / len(chunk),
                         "postprocess": (t3 - t2) * 1e3 / len(chunk)}
                r = Results(orig, paths[i0 + j], self.names, torch.from_numpy(d), m, speed, net_shape)
                if save_dir:
                    r.save_dir = save_dir
                    r.save(os.path.join(save_dir, os.path.splitext(os.path.basename(paths[i0 + j]))[0] + ".jpg"))
                if verbose:
                    print(f"image {i0 + j + 1}/{len(imgs)} {paths[i0 + j]}: {net_shape[0]}x{net_shape[1]} {r.verbose()}"
                          f"{speed['inference']:.1f}ms")
                results.append(r)
        if verbose and save_dir:
            print(f"Results saved to {save_dir}")
        return results

    __call__ = predict

    # ------------------------------------------------------------------ training
    def train(self, data: Optional[str] = None, epochs: int = 100, imgsz: int = 640, batch: int = 16,
              project: Optional[str] = None, name: Optional[str] = None, device=0, **kwargs):
        from .train import train as _train  # lazy: training pulls in the loss / dataset modules
        return _train(self, data=data, epochs=epochs, imgsz=imgsz, batch=batch, project=project, name=name,
                      device=device, **kwargs)

    def val(self, data: Optional[str] = None, imgsz: Optional[int] = None, batch: int = 16, device=0, **kwargs):
        from .train import validate as _validate
        return _validate(self, data=data, imgsz=imgsz, batch=batch, device=device, **kwargs)
