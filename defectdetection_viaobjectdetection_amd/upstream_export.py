"""Writing checkpoints that upstream Ultralytics can load (write side).

The reference reloads what training wrote -- ``YOLO(os.path.join(project, "run/weights/best.pt"))``,
/root/reference/BscanBased/yolo/yolo_eval.py:7-10 -- so a user moving between this build and upstream needs a ``.pt``
whose pickle upstream's ``torch.load`` accepts.  Such a file is ``{'model': <SegmentationModel nn.Module>, 'ema': None,
'train_args': {...}, 'epoch': -1, ...}`` with the module graph pickled BY CLASS REFERENCE
(``ultralytics.nn.tasks.SegmentationModel``, ``ultralytics.nn.modules.conv.Conv``, ...).  This module builds that object
graph out of real ``torch.nn`` leaves (Conv2d, BatchNorm2d, SiLU, Upsample, MaxPool2d, ConvTranspose2d, Sequential,
ModuleList) and thin ``nn.Module`` stand-ins for the upstream classes, registered under upstream's module paths only
while ``torch.save`` runs, carrying the attributes upstream's ``forward`` / ``fuse`` / ``_predict_once`` read
(``Conv.conv/bn/act``, ``C2f.c``, ``Bottleneck.add``, ``Concat.d``, ``Segment.nc/nl/reg_max/no/nm/npr/stride``, the
per-layer ``f / i / type / np`` tags, ``model.yaml / save / names / stride / inplace / args``).  No upstream code is
copied or executed: the stand-ins have no methods; upstream supplies the behaviour when it unpickles into its own classes.

What cannot be verified here: ``ultralytics`` is not installable offline, so the file is checked by (i) the
exact set of pickled globals, (ii) a plain ``torch.load`` with stand-in classes registered, whose rebuilt module graph must
return the original ``state_dict()``, and (iii) a round trip through this package's own reader (``upstream_ckpt.py``) --
tests/test_upstream_export_host.py.
"""
from __future__ import annotations

import contextlib
import sys
import types
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .spec import NM, REG_MAX, SCALES, V9C, _make_divisible

_UP = {"Conv": "ultralytics.nn.modules.conv", "Concat": "ultralytics.nn.modules.conv",
       "C2f": "ultralytics.nn.modules.block", "Bottleneck": "ultralytics.nn.modules.block",
       "SPPF": "ultralytics.nn.modules.block", "Proto": "ultralytics.nn.modules.block", "DFL": "ultralytics.nn.modules.block",
       "Segment": "ultralytics.nn.modules.head", "SegmentationModel": "ultralytics.nn.tasks",
       # yolov9c-seg (GELAN): the blocks /root/reference/BscanBased/yolo_seg_train.py:7 and yolo8_seg_predict.py:4 name
       "RepConv": "ultralytics.nn.modules.conv", "RepBottleneck": "ultralytics.nn.modules.block",
       "RepCSP": "ultralytics.nn.modules.block", "RepNCSPELAN4": "ultralytics.nn.modules.block",
       "ADown": "ultralytics.nn.modules.block", "SPPELAN": "ultralytics.nn.modules.block"}

_classes: Dict[str, type] = {}


def _cls(name: str) -> type:
    """An nn.Module subclass named like the upstream class, living (by __module__) at upstream's module path."""
    if name not in _classes:
        _classes[name] = type(name, (nn.Module,), {"__module__": _UP[name], "__qualname__": name})
    return _classes[name]


@contextlib.contextmanager
def _registered():
    """Make the stand-in classes importable under upstream's paths while pickle resolves them; leave no trace after."""
    created, replaced = [], {}
    try:
        for name, mod in _UP.items():
            parts = mod.split(".")
            for i in range(1, len(parts) + 1):
                mn = ".".join(parts[:i])
                if mn not in sys.modules:
                    sys.modules[mn] = types.ModuleType(mn)
                    created.append(mn)
            m = sys.modules[mod]
            if hasattr(m, name) and getattr(m, name) is not _cls(name):
                replaced[(mod, name)] = getattr(m, name)
            setattr(m, name, _cls(name))
        yield
    finally:
        for (mod, name), old in replaced.items():
            setattr(sys.modules[mod], name, old)
        for mn in created:
            sys.modules.pop(mn, None)


def _conv(c1: int, c2: int, k: int = 1, s: int = 1) -> nn.Module:
    m = _cls("Conv")()
    m.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
    m.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
    m.act = nn.SiLU(inplace=True)
    return m


def _conv_noact(c1: int, c2: int, k: int) -> nn.Module:
    m = _conv(c1, c2, k, 1)
    m.act = nn.Identity()
    return m


def _repconv(c1: int, c2: int) -> nn.Module:
    """upstream's RepConv(c1, c2, 3, 1) as RepBottleneck builds it: a 3x3 and a 1x1 Conv without activation, no identity
    branch (bn=False), SiLU after the sum."""
    m = _cls("RepConv")()
    m.g, m.c1, m.c2 = 1, c1, c2
    m.act = nn.SiLU(inplace=True)
    m.bn = None
    m.conv1 = _conv_noact(c1, c2, 3)
    m.conv2 = _conv_noact(c1, c2, 1)
    return m


def _repbottleneck(c1: int, c2: int) -> nn.Module:
    m = _cls("RepBottleneck")()
    m.cv1 = _repconv(c1, c2)
    m.cv2 = _conv(c2, c2, 3, 1)
    m.add = c1 == c2
    return m


def _repcsp(c1: int, c2: int, n: int = 1) -> nn.Module:
    m = _cls("RepCSP")()
    c_ = c2 // 2
    m.cv1 = _conv(c1, c_, 1, 1)
    m.cv2 = _conv(c1, c_, 1, 1)
    m.cv3 = _conv(2 * c_, c2, 1, 1)
    m.m = nn.Sequential(*(_repbottleneck(c_, c_) for _ in range(n)))
    return m


def _elan(c1: int, c2: int, c3: int, c4: int, n: int = 1) -> nn.Module:
    m = _cls("RepNCSPELAN4")()
    m.c = c3 // 2
    m.cv1 = _conv(c1, c3, 1, 1)
    m.cv2 = nn.Sequential(_repcsp(c3 // 2, c4, n), _conv(c4, c4, 3, 1))
    m.cv3 = nn.Sequential(_repcsp(c4, c4, n), _conv(c4, c4, 3, 1))
    m.cv4 = _conv(c3 + 2 * c4, c2, 1, 1)
    return m


def _adown(c1: int, c2: int) -> nn.Module:
    m = _cls("ADown")()
    m.c = c2 // 2
    m.cv1 = _conv(c1 // 2, m.c, 3, 2)
    m.cv2 = _conv(c1 // 2, m.c, 1, 1)
    return m


def _sppelan(c1: int, c2: int, c3: int, k: int = 5) -> nn.Module:
    m = _cls("SPPELAN")()
    m.c = c3
    m.cv1 = _conv(c1, c3, 1, 1)
    m.cv2 = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
    m.cv3 = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
    m.cv4 = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
    m.cv5 = _conv(4 * c3, c2, 1, 1)
    return m


def _bottleneck(c1: int, c2: int, shortcut: bool) -> nn.Module:
    m = _cls("Bottleneck")()
    m.cv1 = _conv(c1, c2, 3, 1)
    m.cv2 = _conv(c2, c2, 3, 1)
    m.add = bool(shortcut and c1 == c2)
    return m


def _c2f(c1: int, c2: int, n: int, shortcut: bool) -> nn.Module:
    m = _cls("C2f")()
    m.c = c2 // 2
    m.cv1 = _conv(c1, 2 * m.c, 1, 1)
    m.cv2 = _conv((2 + n) * m.c, c2, 1, 1)
    m.m = nn.ModuleList(_bottleneck(m.c, m.c, shortcut) for _ in range(n))
    return m


def _sppf(c1: int, c2: int, k: int = 5) -> nn.Module:
    m = _cls("SPPF")()
    m.cv1 = _conv(c1, c1 // 2, 1, 1)
    m.cv2 = _conv(c1 // 2 * 4, c2, 1, 1)
    m.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
    return m


def _concat() -> nn.Module:
    m = _cls("Concat")()
    m.d = 1
    return m


def _segment(nc: int, nm: int, npr: int, ch) -> nn.Module:
    m = _cls("Segment")()
    m.nc, m.nl, m.reg_max = nc, len(ch), REG_MAX
    m.no = nc + REG_MAX * 4
    m.stride = torch.tensor([8.0, 16.0, 32.0])
    c2, c3, c4 = max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], min(nc, 100)), max(ch[0] // 4, nm)
    m.cv2 = nn.ModuleList(nn.Sequential(_conv(x, c2, 3), _conv(c2, c2, 3), nn.Conv2d(c2, 4 * REG_MAX, 1)) for x in ch)
    m.cv3 = nn.ModuleList(nn.Sequential(_conv(x, c3, 3), _conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
    dfl = _cls("DFL")()
    dfl.conv = nn.Conv2d(REG_MAX, 1, 1, bias=False).requires_grad_(False)
    dfl.c1 = REG_MAX
    m.dfl = dfl
    m.nm, m.npr = nm, npr
    proto = _cls("Proto")()
    proto.cv1 = _conv(ch[0], npr, 3)
    proto.upsample = nn.ConvTranspose2d(npr, npr, 2, 2, 0, bias=True)
    proto.cv2 = _conv(npr, npr, 3)
    proto.cv3 = _conv(npr, nm, 1)
    m.proto = proto
    m.cv4 = nn.ModuleList(nn.Sequential(_conv(x, c4, 3), _conv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch)
    return m


def _yaml(scale: str, nc: int) -> Dict:
    """The model description upstream keeps in ``model.yaml`` (yolov8-seg.yaml)."""
    return {"nc": nc, "scales": {k: list(v) for k, v in SCALES.items()}, "scale": scale, "ch": 3,
            "yaml_file": f"yolov8{scale}-seg.yaml",
            "backbone": [[-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C2f", [128, True]],
                         [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C2f", [256, True]], [-1, 1, "Conv", [512, 3, 2]],
                         [-1, 6, "C2f", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C2f", [1024, True]],
                         [-1, 1, "SPPF", [1024, 5]]],
            "head": [[-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 6], 1, "Concat", [1]], [-1, 3, "C2f", [512]],
                     [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 4], 1, "Concat", [1]], [-1, 3, "C2f", [256]],
                     [-1, 1, "Conv", [256, 3, 2]], [[-1, 12], 1, "Concat", [1]], [-1, 3, "C2f", [512]],
                     [-1, 1, "Conv", [512, 3, 2]], [[-1, 9], 1, "Concat", [1]], [-1, 3, "C2f", [1024]],
                     [[15, 18, 21], 1, "Segment", ["nc", 32, 256]]]}


def _yaml_v9c(nc: int) -> Dict:
    """yolov9c-seg.yaml (the file /root/reference/BscanBased/yolo_seg_train.py:7 names): GELAN-C backbone + head."""
    e = "RepNCSPELAN4"
    return {"nc": nc, "ch": 3, "yaml_file": "yolov9c-seg.yaml",
            "backbone": [[-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 1, e, [256, 128, 64, 1]],
                         [-1, 1, "ADown", [256]], [-1, 1, e, [512, 256, 128, 1]], [-1, 1, "ADown", [512]],
                         [-1, 1, e, [512, 512, 256, 1]], [-1, 1, "ADown", [512]], [-1, 1, e, [512, 512, 256, 1]],
                         [-1, 1, "SPPELAN", [512, 256]]],
            "head": [[-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 6], 1, "Concat", [1]], [-1, 1, e, [512, 512, 256, 1]],
                     [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 4], 1, "Concat", [1]], [-1, 1, e, [256, 256, 128, 1]],
                     [-1, 1, "ADown", [256]], [[-1, 12], 1, "Concat", [1]], [-1, 1, e, [512, 512, 256, 1]],
                     [-1, 1, "ADown", [512]], [[-1, 9], 1, "Concat", [1]], [-1, 1, e, [512, 512, 256, 1]],
                     [[15, 18, 21], 1, "Segment", ["nc", 32, 256]]]}


def _finish_model(layers: List[nn.Module], froms, yaml: Dict, nc: int, names: Dict[int, str],
                  state_dict: Dict[str, torch.Tensor], train_args: Optional[Dict]) -> nn.Module:
    """Tag the layers the way upstream's parse_model does (``i / f / type / np``), wrap them as ``SegmentationModel`` and
    load ``state_dict`` (upstream key names, strict)."""
    for i, (m, f) in enumerate(zip(layers, froms)):
        m.i, m.f = i, f
        name = type(m).__name__
        m.type = "torch.nn.modules.upsampling.Upsample" if name == "Upsample" else f"{_UP[name]}.{name}"
        m.np = sum(p.numel() for p in m.parameters())
    model = _cls("SegmentationModel")()
    model.yaml = yaml
    model.model = nn.Sequential(*layers)
    model.save = [4, 6, 9, 12, 15, 18, 21]           # layers whose output a later layer reads (sorted)
    model.names = {int(k): str(v) for k, v in names.items()}
    model.inplace = True
    model.stride = torch.tensor([8.0, 16.0, 32.0])
    model.nc = nc
    model.args = dict(train_args or {})
    model.task = "segment"
    want = OrderedDict((k, v) for k, v in model.state_dict().items())
    missing = [k for k in want if k not in state_dict]
    if missing:
        raise KeyError(f"state dict misses {len(missing)} upstream keys, e.g. {missing[:3]}")
    model.load_state_dict({k: state_dict[k] for k in want}, strict=True)
    return model.eval()


def build_upstream_module_v9c(nc: int, names: Dict[int, str], state_dict: Dict[str, torch.Tensor],
                              train_args: Optional[Dict] = None) -> nn.Module:
    """The yolov9c-seg ``SegmentationModel`` stand-in graph: the same 23 entries, same froms and same saved layers as the
    yolov8-seg graph, GELAN blocks in place of C2f / SPPF and ADown in place of the stride-2 convs."""
    layers: List[nn.Module] = [
        _conv(3, 64, 3, 2), _conv(64, 128, 3, 2), _elan(128, 256, 128, 64), _adown(256, 256), _elan(256, 512, 256, 128),
        _adown(512, 512), _elan(512, 512, 512, 256), _adown(512, 512), _elan(512, 512, 512, 256), _sppelan(512, 512, 256),
        nn.Upsample(None, 2, "nearest"), _concat(), _elan(1024, 512, 512, 256),
        nn.Upsample(None, 2, "nearest"), _concat(), _elan(1024, 256, 256, 128),
        _adown(256, 256), _concat(), _elan(768, 512, 512, 256),
        _adown(512, 512), _concat(), _elan(1024, 512, 512, 256),
        _segment(nc, NM, 256, (256, 512, 512)),
    ]
    froms = [-1] * 10 + [-1, [-1, 6], -1, -1, [-1, 4], -1, -1, [-1, 12], -1, -1, [-1, 9], -1, [15, 18, 21]]
    return _finish_model(layers, froms, _yaml_v9c(nc), nc, names, state_dict, train_args)


def build_upstream_module(scale: str, nc: int, names: Dict[int, str], state_dict: Dict[str, torch.Tensor],
                          train_args: Optional[Dict] = None) -> nn.Module:
    """The ``SegmentationModel`` stand-in graph holding ``state_dict`` (upstream key names)."""
    if scale == V9C:
        return build_upstream_module_v9c(nc, names, state_dict, train_args)
    depth, width, maxc = SCALES[scale]
    ch = lambda c: _make_divisible(min(c, maxc) * width, 8)  # noqa: E731
    rep = lambda n: max(round(n * depth), 1) if n > 1 else n  # noqa: E731
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    layers: List[nn.Module] = [
        _conv(3, c64, 3, 2), _conv(c64, c128, 3, 2), _c2f(c128, c128, rep(3), True), _conv(c128, c256, 3, 2),
        _c2f(c256, c256, rep(6), True), _conv(c256, c512, 3, 2), _c2f(c512, c512, rep(6), True), _conv(c512, c1024, 3, 2),
        _c2f(c1024, c1024, rep(3), True), _sppf(c1024, c1024, 5),
        nn.Upsample(None, 2, "nearest"), _concat(), _c2f(c1024 + c512, c512, rep(3), False),
        nn.Upsample(None, 2, "nearest"), _concat(), _c2f(c512 + c256, c256, rep(3), False),
        _conv(c256, c256, 3, 2), _concat(), _c2f(c256 + c512, c512, rep(3), False),
        _conv(c512, c512, 3, 2), _concat(), _c2f(c512 + c1024, c1024, rep(3), False),
        _segment(nc, NM, ch(256), (c256, c512, c1024)),
    ]
    froms = [-1] * 10 + [-1, [-1, 6], -1, -1, [-1, 4], -1, -1, [-1, 12], -1, -1, [-1, 9], -1, [15, 18, 21]]
    return _finish_model(layers, froms, _yaml(scale, nc), nc, names, state_dict, train_args)


def export_upstream_checkpoint(path: str, scale: str, nc: int, names: Dict[int, str], state_dict: Dict[str, torch.Tensor],
                               train_args: Optional[Dict] = None, half: bool = True) -> str:
    """Write ``path`` in upstream's checkpoint layout (``best.pt``: EMA weights as fp16 module, no optimizer)."""
    import datetime
    targs = {"task": "segment", "mode": "train", "imgsz": 640, **(train_args or {})}
    with _registered():
        model = build_upstream_module(scale, nc, names, state_dict, targs)
        if half:
            model = model.half()
        ckpt = {"epoch": -1, "best_fitness": None, "model": model, "ema": None, "updates": None, "optimizer": None,
                "train_args": targs, "train_metrics": {}, "train_results": {}, "date": datetime.datetime.now().isoformat(),
                "version": "8.3.0", "license": "AGPL-3.0 (https://ultralytics.com/license)", "docs": "https://docs.ultralytics.com"}
        torch.save(ckpt, path)
    return path
