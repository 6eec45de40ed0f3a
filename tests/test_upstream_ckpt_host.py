"""Loading an upstream-style pickled checkpoint when the `ultralytics` classes cannot be imported (row N3).
The fixture is fabricated here: the oracle's modules are pickled under upstream's module paths, which are then
removed from sys.modules -- exactly the situation of a user's best.pt on a machine without the package.  CPU only."""
import os
import pickle
import sys
import types

import pytest
import torch

import yolov8_seg_oracle as orc
from defectdetection_viaobjectdetection_amd.model import YOLO
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
from defectdetection_viaobjectdetection_amd.upstream_ckpt import load_upstream_checkpoint

FAKE = {"Conv": "ultralytics.nn.modules.conv", "Bottleneck": "ultralytics.nn.modules.block", "C2f": "ultralytics.nn.modules.block",
        "SPPF": "ultralytics.nn.modules.block", "Proto": "ultralytics.nn.modules.block", "DFL": "ultralytics.nn.modules.block",
        "Segment": "ultralytics.nn.modules.head", "SegmentationModel": "ultralytics.nn.tasks"}


def _write_fake_upstream_ckpt(path, scale, nc, half=True, as_ema=False):
    sd = synthetic_state_dict(scale, nc, seed=4)
    model = orc.SegmentationModel(scale, nc)
    model.load_state_dict(sd)
    model.names = {i: f"defect{i}" for i in range(nc)}
    saved_mod, created = {}, []
    try:
        for cls_name, mod_name in FAKE.items():
            cls = getattr(orc, cls_name)
            saved_mod[cls] = (cls.__module__, cls.__qualname__)
            parts = mod_name.split(".")
            for i in range(1, len(parts) + 1):
                mn = ".".join(parts[:i])
                if mn not in sys.modules:
                    sys.modules[mn] = types.ModuleType(mn)
                    created.append(mn)
            setattr(sys.modules[mod_name], cls_name, cls)
            cls.__module__ = mod_name
        m = model.half() if half else model
        ck = {"epoch": 29, "best_fitness": None, "model": None if as_ema else m, "ema": m if as_ema else None, "updates": 100,
              "optimizer": None, "train_args": {"imgsz": 320, "epochs": 30, "data": "data-seg.yaml", "device": 0},
              "date": "2025-01-01", "version": "8.3.0"}
        torch.save(ck, path)
    finally:
        for cls, (mn, qn) in saved_mod.items():
            cls.__module__ = mn
        for mn in created:
            sys.modules.pop(mn, None)
    return sd


@pytest.mark.parametrize("scale,nc,half,as_ema", [("n", 1, True, False), ("s", 3, False, True)])
def test_upstream_style_checkpoint_loads_without_its_classes(tmp_path, scale, nc, half, as_ema):
    path = str(tmp_path / "best.pt")
    sd = _write_fake_upstream_ckpt(path, scale, nc, half, as_ema)
    assert "ultralytics.nn.tasks" not in sys.modules            # the pickled classes are really gone
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)  # plain torch.load needs them
    up = load_upstream_checkpoint(path)
    assert up["scale"] == scale and up["nc"] == nc and up["names"] == {i: f"defect{i}" for i in range(nc)}
    assert up["train_args"]["imgsz"] == 320
    for k, v in sd.items():
        ref = v.half().float() if (half and v.is_floating_point()) else v
        assert torch.equal(up["state_dict"][k].float() if v.is_floating_point() else up["state_dict"][k], ref.float() if v.is_floating_point() else ref), k
    m = YOLO(path)                                               # the reference's call: YOLO(model_path)
    assert m.scale == scale and m.nc == nc and m.train_args["imgsz"] == 320 and len(m.state_dict) == len(sd)


def test_rejects_other_graphs_and_dangerous_pickles(tmp_path):
    p = str(tmp_path / "other.pt")
    torch.save({"model": torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3)), "train_args": {}}, p)
    with pytest.raises(ValueError, match="not a YOLOv8-seg graph"):
        load_upstream_checkpoint(p)

    class Evil:
        def __reduce__(self):
            return (eval, ("1+1",))
    p2 = str(tmp_path / "evil.pt")
    torch.save({"model": Evil()}, p2)
    with pytest.raises(pickle.UnpicklingError):
        load_upstream_checkpoint(p2)


def _raw_pickle(module, name, arg):
    """protocol-4 pickle: STACK_GLOBAL(module, name) called with (arg,) -- what a hostile checkpoint would carry."""
    def s(x):
        b = x.encode()
        return b"\x8c" + bytes([len(b)]) + b
    return b"\x80\x04" + s(module) + s(name) + b"\x93" + s(arg) + b"\x85R."


@pytest.mark.parametrize("module,name", [
    ("torch.serialization", "os.system"),                # dotted attribute path out of an allowed package (protocol 4)
    ("torch.serialization", "os.getpid"),
    ("types", "FunctionType"), ("types", "CodeType"),
    ("numpy.testing._private.utils", "runstring"),       # calls exec
    ("torch.storage", "_load_from_bytes"),               # torch.load(weights_only=False) on attacker bytes
    ("torch.serialization", "load"), ("torch.hub", "load"), ("torch", "load"),
    ("builtins", "eval"), ("builtins", "exec"), ("builtins", "getattr"), ("builtins", "__import__"),
    ("os", "system"), ("posix", "system"), ("subprocess", "Popen"), ("functools", "partial"), ("operator", "attrgetter"),
])
def test_unpickler_refuses_everything_off_the_allow_list(module, name):
    import io
    from defectdetection_viaobjectdetection_amd.upstream_ckpt import _Unpickler
    with pytest.raises(pickle.UnpicklingError):
        out = _Unpickler(io.BytesIO(_raw_pickle(module, name, "echo pwned"))).load()
        raise AssertionError(f"{module}.{name} resolved and returned {out!r}")


def test_model_loader_only_falls_back_on_unpickling_errors(tmp_path):
    with pytest.raises(Exception) as ei:
        p = tmp_path / "garbage.pt"
        p.write_bytes(b"not a checkpoint at all")
        YOLO(str(p))
    assert not isinstance(ei.value, (AttributeError, NotImplementedError))
