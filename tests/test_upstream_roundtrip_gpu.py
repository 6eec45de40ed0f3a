"""Upstream-layout checkpoint -> ``YOLO(path)`` -> predict on the GPU (SURVEY next row N3, both directions, both graphs):
``YOLO.save(path, upstream=True)`` writes upstream Ultralytics' pickled-module layout (fp16 weights, class references
``ultralytics.nn.tasks.SegmentationModel`` ...), ``YOLO(path)`` reads it back without the package
(/root/reference/BscanBased/yolo8_seg_predict.py:4-5 loads such a file, a yolov9c-seg one), and the detections of the
reloaded model are BIT-IDENTICAL to the source model's -- for source weights that fp16 represents exactly (upstream stores
``model.half()``; a weight that is not an fp16 value cannot survive any upstream checkpoint unchanged)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("arch,scale,cls_bias,seed", [("yolov9c-seg", "9c", -2.5, 2), ("yolov8n-seg", "n", -2.0, 0)])
def test_upstream_layout_file_predicts_like_its_source(tmp_path, cuda_device, arch, scale, cls_bias, seed):
    from ultralytics import YOLO                      # the shim
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    sd = synthetic_state_dict(scale, 1, seed=seed, cls_bias=cls_bias)
    sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}     # what a .half() checkpoint can hold
    src = YOLO(f"{arch}.yaml")
    src.set_classes(1, {0: "defect"})
    src.load_state_dict(sd)
    src.train_args = {"imgsz": 640}
    path = src.save(str(tmp_path / arch / "segmentation320" / "weights" / "best.pt"), upstream=True)
    with pytest.raises(Exception):                    # it IS upstream's layout: plain torch.load needs upstream's classes
        torch.load(path, map_location="cpu", weights_only=False)
    model = YOLO(path)                                # yolo8_seg_predict.py:5
    assert model.scale == scale and model.nc == 1 and model.names == {0: "defect"} and model.train_args["imgsz"] == 640
    for k, v in sd.items():
        assert torch.equal(model.state_dict[k], v), k
    png = os.path.join(GOLDEN, "bscans", "787-225_01_Ch-0_51.png")
    a = src.predict(png, verbose=False, conf=0.25)[0]             # imgsz from the checkpoint's train args (D7)
    b = model.predict(png, save=True, project=str(tmp_path / "runs"), name="predict", verbose=False, conf=0.25)[0]   # :8
    assert len(b.boxes) > 0, "the synthetic weights must detect something on the fixture"
    assert np.array_equal(a.boxes.data.numpy(), b.boxes.data.numpy())
    assert torch.equal(a.masks.data, b.masks.data)
    assert b.orig_shape == (320, 320) and os.path.isfile(os.path.join(b.save_dir, "787-225_01_Ch-0_51.jpg"))
