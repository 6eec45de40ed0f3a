"""The upstream-loadable exporter (row N3, write side): `YOLO.save(path, upstream=True)`.  CPU only.
`ultralytics` cannot be installed here, so the file is checked by what it CONTAINS and by two independent
loaders: plain torch.load with stand-in classes at upstream's module paths, and this package's class-free reader."""
import io
import pickle
import pickletools
import sys
import zipfile

import pytest
import torch
import torch.nn as nn

from defectdetection_viaobjectdetection_amd.model import YOLO
from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
from defectdetection_viaobjectdetection_amd.upstream_ckpt import load_upstream_checkpoint
from defectdetection_viaobjectdetection_amd.upstream_export import _registered


def _globals_of(path):
    with zipfile.ZipFile(path) as z:
        name = [n for n in z.namelist() if n.endswith("data.pkl")][0]
        data = z.read(name)
    out, strs = set(), []
    for op, arg, _ in pickletools.genops(io.BytesIO(data)):
        if op.name == "GLOBAL":
            out.add(tuple(arg.split(" ")))
        elif op.name in ("BINUNICODE", "SHORT_BINUNICODE", "UNICODE"):
            strs.append(arg)
        elif op.name == "STACK_GLOBAL":
            out.add((strs[-2], strs[-1]))
    return out


@pytest.mark.parametrize("scale,nc", [("n", 1), ("s", 3)])
def test_export_is_loadable_by_class_reference_and_round_trips(tmp_path, scale, nc):
    sd = synthetic_state_dict(scale, nc, seed=5)
    m = YOLO(f"yolov8{scale}-seg.yaml")
    m.set_classes(nc, {i: f"defect{i}" for i in range(nc)})
    m.load_state_dict(sd)
    m.train_args = {"imgsz": 320, "data": "data-seg.yaml", "epochs": 30}
    path = str(tmp_path / "best.pt")
    m.save(path, upstream=True)
    assert "ultralytics.nn.tasks" not in sys.modules          # the exporter left nothing registered
    # (i) the pickle references exactly upstream's class paths (+ torch / collections), nothing of this package
    g = _globals_of(path)
    mods = {a for a, _ in g}
    assert ("ultralytics.nn.tasks", "SegmentationModel") in g and ("ultralytics.nn.modules.head", "Segment") in g
    for cls, mod in (("Conv", "ultralytics.nn.modules.conv"), ("Concat", "ultralytics.nn.modules.conv"), ("C2f", "ultralytics.nn.modules.block"),
                     ("Bottleneck", "ultralytics.nn.modules.block"), ("SPPF", "ultralytics.nn.modules.block"),
                     ("Proto", "ultralytics.nn.modules.block"), ("DFL", "ultralytics.nn.modules.block")):
        assert (mod, cls) in g, (mod, cls)
    assert not any(x.startswith("defectdetection") or x.startswith("oracle") or x == "__main__" for x in mods), mods
    assert all(x.split(".")[0] in ("ultralytics", "torch", "collections", "builtins", "__builtin__", "_codecs", "numpy") for x in mods), mods
    # (ii) plain torch.load, stand-in classes importable at upstream's paths: the module graph comes back with its tensors
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)          # without the classes it cannot load
    with _registered():
        ck = torch.load(path, map_location="cpu", weights_only=False)
    model = ck["model"]
    assert isinstance(model, nn.Module) and type(model).__name__ == "SegmentationModel" and ck["ema"] is None
    assert ck["train_args"]["imgsz"] == 320 and model.names == {i: f"defect{i}" for i in range(nc)}
    got = model.state_dict()
    assert set(got) == set(sd)            # (upstream's registration order puts C2f.cv2 before C2f.m: the order differs, the keys do not)
    for k, v in sd.items():
        ref = v.half() if v.is_floating_point() else v
        assert torch.equal(got[k], ref), k
    # attributes upstream's forward / fuse / _predict_once read
    seq = model.model
    assert [getattr(l, "f") for l in seq][10:] == [-1, [-1, 6], -1, -1, [-1, 4], -1, -1, [-1, 12], -1, -1, [-1, 9], -1, [15, 18, 21]]
    assert [l.i for l in seq] == list(range(23)) and model.save == [4, 6, 9, 12, 15, 18, 21]
    assert model.yaml["head"][-1][2] == "Segment" and model.yaml["scale"] == scale and model.yaml["nc"] == nc
    head = seq[22]
    assert (head.nc, head.nl, head.reg_max, head.no, head.nm) == (nc, 3, 16, nc + 64, 32) and head.stride.tolist() == [8.0, 16.0, 32.0]
    assert isinstance(seq[0].conv, nn.Conv2d) and isinstance(seq[0].bn, nn.BatchNorm2d) and isinstance(seq[0].act, nn.SiLU)
    assert seq[2].c == seq[2].cv2.conv.out_channels // 2 and seq[2].m[0].add is True and seq[12].m[0].add is False
    assert seq[11].d == 1 and isinstance(seq[9].m, nn.MaxPool2d) and isinstance(seq[10], nn.Upsample)
    # (iii) this package's class-free reader recovers the same checkpoint
    up = load_upstream_checkpoint(path)
    assert up["scale"] == scale and up["nc"] == nc and up["train_args"]["imgsz"] == 320
    for k, v in sd.items():
        ref = v.half().float() if v.is_floating_point() else v
        assert torch.equal(up["state_dict"][k].float() if v.is_floating_point() else up["state_dict"][k], ref), k
    again = YOLO(path)
    assert again.scale == scale and again.nc == nc and again.names == {i: f"defect{i}" for i in range(nc)}


def test_v9c_export_is_loadable_by_class_reference_and_round_trips(tmp_path):
    """The checkpoint /root/reference/BscanBased/yolo8_seg_predict.py:4-5 loads is a yolov9c-seg one: the exporter writes that
    graph too (GELAN stand-ins RepNCSPELAN4 / RepCSP / RepBottleneck / RepConv / ADown / SPPELAN at upstream's paths)."""
    sd = synthetic_state_dict("9c", 2, seed=4)
    m = YOLO("yolov9c-seg.yaml")
    m.set_classes(2, {0: "defect", 1: "porosity"})
    m.load_state_dict(sd)
    m.train_args = {"imgsz": 320, "data": "data-seg.yaml", "epochs": 30}
    path = str(tmp_path / "yolo9c-seg" / "segmentation320" / "weights" / "best.pt")
    m.save(path, upstream=True)
    assert "ultralytics.nn.tasks" not in sys.modules
    g = _globals_of(path)
    for cls, mod in (("SegmentationModel", "ultralytics.nn.tasks"), ("Segment", "ultralytics.nn.modules.head"),
                     ("Conv", "ultralytics.nn.modules.conv"), ("RepConv", "ultralytics.nn.modules.conv"),
                     ("Concat", "ultralytics.nn.modules.conv"), ("RepNCSPELAN4", "ultralytics.nn.modules.block"),
                     ("RepCSP", "ultralytics.nn.modules.block"), ("RepBottleneck", "ultralytics.nn.modules.block"),
                     ("ADown", "ultralytics.nn.modules.block"), ("SPPELAN", "ultralytics.nn.modules.block"),
                     ("Proto", "ultralytics.nn.modules.block"), ("DFL", "ultralytics.nn.modules.block")):
        assert (mod, cls) in g, (mod, cls)
    mods = {a for a, _ in g}
    assert ("ultralytics.nn.modules.block", "C2f") not in g and ("ultralytics.nn.modules.block", "SPPF") not in g
    assert not any(x.startswith("defectdetection") or x.startswith("oracle") or x == "__main__" for x in mods), mods
    with _registered():
        ck = torch.load(path, map_location="cpu", weights_only=False)
    model = ck["model"]
    assert type(model).__name__ == "SegmentationModel" and model.yaml["yaml_file"] == "yolov9c-seg.yaml"
    got = model.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v.half() if v.is_floating_point() else v), k
    seq = model.model
    assert [type(l).__name__ for l in seq][:10] == ["Conv", "Conv", "RepNCSPELAN4", "ADown", "RepNCSPELAN4", "ADown", "RepNCSPELAN4",
                                                    "ADown", "RepNCSPELAN4", "SPPELAN"]
    assert [l.i for l in seq] == list(range(23)) and model.save == [4, 6, 9, 12, 15, 18, 21]
    rep = seq[2].cv2[0].m[0].cv1                      # RepConv: two activation-free branches, SiLU after their sum, no identity branch
    assert type(rep).__name__ == "RepConv" and rep.bn is None and isinstance(rep.act, nn.SiLU)
    assert isinstance(rep.conv1.act, nn.Identity) and rep.conv1.conv.kernel_size == (3, 3) and rep.conv2.conv.kernel_size == (1, 1)
    assert seq[2].c == 64 and seq[3].c == 128 and seq[9].c == 256 and isinstance(seq[9].cv3, nn.MaxPool2d)
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for k, v in sd.items()
                                                            if "running_" not in k and "num_batches" not in k)
    up = load_upstream_checkpoint(path)
    assert up["scale"] == "9c" and up["nc"] == 2 and up["names"] == {0: "defect", 1: "porosity"}
    for k, v in sd.items():
        ref = v.half().float() if v.is_floating_point() else v
        assert torch.equal(up["state_dict"][k].float() if v.is_floating_point() else up["state_dict"][k], ref), k
    again = YOLO(path)                                  # /root/reference/BscanBased/yolo8_seg_predict.py:5
    assert again.scale == "9c" and again.nc == 2
