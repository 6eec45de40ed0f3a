"""Reading checkpoints written by upstream Ultralytics WITHOUT the ultralytics package.

The reference loads weights by path (`YOLO(model_path)`, /root/reference/BscanBased/yolo8_seg_predict.py:5-6); such a
`.pt` is a pickle of `{'model': <SegmentationModel nn.Module>, 'ema': ..., 'train_args': {...}, ...}` whose class
definitions live in the un-vendored package.  Nothing of those classes is needed to recover the weights: this module
unpickles with an EXACT allow-list (torch's tensor rebuild functions, storages, dtypes, a few containers and numpy
scalars are real; classes of ultralytics / torch.nn / pathlib / argparse become inert placeholders that only store
their state; every other global, and every dotted attribute path, raises UnpicklingError), then walks the placeholder graph along nn.Module's own
`_modules / _parameters / _buffers` dictionaries and rebuilds the `state_dict()` the model would have produced.  The
graph (YOLOv8{n,s,m,l,x}-seg) is recognised from tensor names and shapes; anything else is rejected with a message.
No code from the pickle is executed beyond torch's tensor rebuild functions.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

# Exact (module, name) pairs that resolve to REAL objects.  Everything a YOLO checkpoint needs to rebuild its tensors and
# plain containers, and nothing that can call back into Python: no attribute paths (pickle protocol 4 resolves dotted
# names), no `torch.serialization`, `torch.storage._load_from_bytes`, `types`, `functools`, `operator`, `os`.
_TORCH_STORAGES = ("FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage",
                   "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage")
_REAL = {("collections", "OrderedDict"), ("collections", "defaultdict"),
         ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
         ("torch._utils", "_rebuild_parameter"), ("torch", "Size"), ("torch", "device"),
         ("torch.storage", "UntypedStorage"),
         ("_codecs", "encode"),
         ("numpy", "dtype"), ("numpy", "ndarray"),
         ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
         ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")}
_REAL |= {("torch", n) for n in _TORCH_STORAGES}
_REAL_BUILTINS = {"set", "frozenset", "slice", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes",
                  "bytearray", "complex", "range", "object"}
# Module roots whose classes are expected in an upstream checkpoint but are never needed as real objects: instances
# become inert placeholders that only keep their state (nn.Module graphs, Path / Namespace values in train_args, ...).
_PLACEHOLDER_ROOTS = {"ultralytics", "models", "utils", "yolov5", "__main__", "pathlib", "argparse", "datetime"}
_PLACEHOLDER_EXACT = {("types", "SimpleNamespace"), ("torch", "Tensor")}


class _Placeholder:
    """Stands for an instance of a class that cannot (or must not) be imported: keeps whatever state the pickle restores."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], (dict, type(None))):
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self.__dict__["_state"] = state

    def __call__(self, *args, **kwargs):     # a placeholder used as a reduce callable yields another placeholder
        return _Placeholder()


_placeholder_types: Dict[Tuple[str, str], type] = {}


def _placeholder_class(module: str, name: str) -> type:
    key = (module, name)
    if key not in _placeholder_types:
        _placeholder_types[key] = type(name.split(".")[-1], (_Placeholder,), {"__module__": module, "_upstream": f"{module}.{name}"})
    return _placeholder_types[key]


def _safe_rebuild_from_type_v2(func, new_type, args, state):
    """torch._tensor._rebuild_from_type_v2 without `new_type.__setstate__` / attribute injection: the tensor only."""
    if func not in (torch._utils._rebuild_tensor_v2, torch._utils._rebuild_tensor, torch._utils._rebuild_parameter):
        raise pickle.UnpicklingError("refusing a tensor rebuild through an unknown function")
    return func(*args)


def _safe_rebuild_parameter_with_state(data, requires_grad, backward_hooks, state):
    return torch._utils._rebuild_parameter(data, requires_grad, backward_hooks)


def _safe_reconstructor(cls, base, state):
    """copyreg._reconstructor for protocol-0/1 pickles: only placeholder classes are ever instantiated."""
    if isinstance(cls, type) and issubclass(cls, _Placeholder):
        return cls()
    raise pickle.UnpicklingError(f"refusing copyreg._reconstructor for {cls!r}")


_SAFE_SUBSTITUTES = {("torch._tensor", "_rebuild_from_type_v2"): _safe_rebuild_from_type_v2,
                     ("torch._utils", "_rebuild_parameter_with_state"): _safe_rebuild_parameter_with_state,
                     ("copyreg", "_reconstructor"): _safe_reconstructor, ("copy_reg", "_reconstructor"): _safe_reconstructor}


class _Unpickler(pickle.Unpickler):
    """Exact allow-list.  (module, name) resolves to a real object only when listed above; names containing '.' never
    do (protocol 4's STACK_GLOBAL would walk attributes: ('torch.serialization', 'os.system') reaches os).  Classes of
    the expected foreign packages become placeholders; anything else is refused."""

    def find_class(self, module, name):
        key = (module, name)
        root = module.split(".")[0]
        if "." in name:
            if root in _PLACEHOLDER_ROOTS:
                return _placeholder_class(module, name)
            raise pickle.UnpicklingError(f"refusing dotted global {module}:{name} in a checkpoint")
        if key in _SAFE_SUBSTITUTES:
            return _SAFE_SUBSTITUTES[key]
        if key in _REAL:
            import importlib
            try:
                return getattr(importlib.import_module(module), name)
            except (ImportError, AttributeError):
                raise pickle.UnpicklingError(f"{module}.{name} is allow-listed but not importable here")
        if root in ("builtins", "__builtin__"):
            if name in _REAL_BUILTINS:
                import builtins
                return getattr(builtins, name)
            raise pickle.UnpicklingError(f"refusing builtins.{name} in a checkpoint")
        if module == "torch" and isinstance(getattr(torch, name, None), torch.dtype):
            return getattr(torch, name)
        if root in _PLACEHOLDER_ROOTS or key in _PLACEHOLDER_EXACT or module.startswith("torch.nn."):
            return _placeholder_class(module, name)
        raise pickle.UnpicklingError(f"refusing global {module}.{name}: not on the checkpoint allow-list")


class _PickleModule:
    """The `pickle_module` torch.load expects: same surface as `pickle`, with the allow-listed Unpickler."""
    Unpickler = _Unpickler
    __name__ = "mi355yolo_upstream_pickle"
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def _walk(obj, prefix: str, out: "OrderedDict[str, torch.Tensor]", seen: set) -> None:
    if id(obj) in seen:
        return
    seen.add(id(obj))
    d = getattr(obj, "__dict__", {})
    for name, p in (d.get("_parameters") or {}).items():
        if isinstance(p, torch.Tensor):
            out[prefix + name] = p.detach()
    for name, b in (d.get("_buffers") or {}).items():
        if isinstance(b, torch.Tensor) and name not in (d.get("_non_persistent_buffers_set") or ()):
            out[prefix + name] = b.detach()
    for name, m in (d.get("_modules") or {}).items():
        if m is not None:
            _walk(m, f"{prefix}{name}.", out, seen)


def module_state_dict(module_like) -> "OrderedDict[str, torch.Tensor]":
    """state_dict() of a (placeholder or real) nn.Module object graph."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _walk(module_like, "", out, set())
    return out


def load_upstream_checkpoint(path: str) -> Dict:
    """Returns {'state_dict', 'scale', 'nc', 'names', 'train_args'} of an upstream YOLOv8-seg checkpoint."""
    from .spec import SCALES, conv_specs, state_dict_keys
    ck = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if not isinstance(ck, dict):
        raise ValueError(f"{path}: expected a checkpoint dict, got {type(ck).__name__}")
    model = ck.get("ema") or ck.get("model")
    if model is None:
        raise ValueError(f"{path}: no 'model' / 'ema' entry")
    sd = model if isinstance(model, dict) else module_state_dict(model)
    sd = OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items())
    stem = sd.get("model.0.conv.weight")
    cls0 = sd.get("model.22.cv3.0.2.weight")
    is_v9c = "model.2.cv2.0.m.0.cv1.conv1.conv.weight" in sd and "model.9.cv5.conv.weight" in sd    # GELAN blocks
    if stem is None or cls0 is None or "model.22.proto.cv1.conv.weight" not in sd:
        kind = getattr(type(model), "_upstream", type(model).__name__)
        raise ValueError(f"{path}: not a YOLOv8-seg graph ({kind}); only yolov8{{n,s,m,l,x}}-seg is implemented "
                         "(SURVEY.md next row N4 lists yolov9c-seg / yolov5u / yolo11)")
    width = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}.get(int(stem.shape[0]))
    if is_v9c:
        from .spec import V9C
        width = V9C
    if width is None or (width not in SCALES and not is_v9c):
        raise ValueError(f"{path}: stem width {int(stem.shape[0])} does not belong to a YOLOv8 scale")
    nc = int(cls0.shape[0])
    keys = state_dict_keys(width, nc)
    missing = [k for k in keys if k not in sd and not k.endswith("num_batches_tracked") and k != "model.22.dfl.conv.weight"]
    if missing:
        raise ValueError(f"{path}: {len(missing)} tensors of yolov8{width}-seg are missing, e.g. {missing[:3]}")
    for s in conv_specs(width, nc):
        if s.rep:
            continue
        k = f"{s.name}.conv.weight" if s.has_bn else f"{s.name}.weight"
        if tuple(sd[k].shape) != s.weight_shape:
            raise ValueError(f"{path}: {k} has shape {tuple(sd[k].shape)}, yolov8{width}-seg expects {s.weight_shape}")
    out = OrderedDict()
    for k in keys:
        if k in sd:
            out[k] = sd[k].clone()
        elif k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        else:
            out[k] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
    names = getattr(model, "names", None) or ck.get("names")
    if isinstance(names, (list, tuple)):
        names = {i: n for i, n in enumerate(names)}
    if not isinstance(names, dict) or len(names) != nc:
        names = {i: f"class{i}" for i in range(nc)}
    targs = ck.get("train_args")
    if not isinstance(targs, dict):
        targs = getattr(targs, "__dict__", {}) if targs is not None else {}
    keep = {k: v for k, v in targs.items() if isinstance(v, (int, float, str, bool, type(None)))}
    return {"state_dict": out, "scale": width, "nc": nc, "names": {int(k): str(v) for k, v in names.items()},
            "train_args": keep}
