"""The reference scripts' call pattern through the ``ultralytics`` shim on the HIP path, checked against the
CPU oracle end to end (config 1 of BASELINE.json: YOLOv8n-seg, one B-scan PNG)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_predict_like_yolo8_seg_predict(tmp_path, cuda_device):
    import yolov8_seg_oracle as orc
    from ultralytics import YOLO  # the shim
    from defectdetection_viaobjectdetection_amd.preprocess import load_image
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict

    model = YOLO("yolov8n-seg.yaml")
    model.set_classes(1, {0: "defect"})                          # data-seg.yaml:4-5
    sd = synthetic_state_dict("n", 1, seed=0, cls_bias=-2.0)
    model.load_state_dict(sd)
    ck = model.save(str(tmp_path / "yolo8n-seg" / "segmentation" / "weights" / "best.pt"))
    model = YOLO(ck)                                             # yolo8_seg_predict.py:5
    png = os.path.join(GOLDEN, "bscans", "787-225_01_Ch-0_51.png")
    results = model.predict(png, save=True, project=str(tmp_path / "runs"), name="predict", conf=0.25)   # :8
    print(results)                                               # :9
    assert len(results) == 1
    res = results[0]
    assert res.orig_shape == (320, 320) and res.names == {0: "defect"}
    assert os.path.isfile(os.path.join(res.save_dir, "787-225_01_Ch-0_51.jpg"))
    assert res.masks is not None and res.masks.data.shape[1:] == (640, 640)
    # oracle end to end on the same image / weights
    omodel = orc.SegmentationModel("n", 1)
    omodel.load_state_dict(sd)
    img = load_image(png)[:, :, ::-1]
    ref = orc.predict(omodel, [np.ascontiguousarray(img)], 640, 0.25, 0.7, 300)[0]
    got = res.boxes.data.numpy()
    # margin rule: detections whose score is within 2e-3 of conf may differ; the rest must match
    strong_ref = ref["boxes"][ref["boxes"][:, 4] > 0.25 + 2e-3]
    strong_got = got[got[:, 4] > 0.25 + 2e-3]
    assert abs(len(strong_ref) - len(strong_got)) <= 1
    n = min(len(strong_ref), len(strong_got), 5)
    assert n > 0, "synthetic weights with cls_bias=-2 must detect something on the fixture"
    assert np.abs(strong_got[:n, :4] - strong_ref[:n, :4]).max() <= 1.0      # original-image pixels
    assert np.abs(strong_got[:n, 4] - strong_ref[:n, 4]).max() <= 5e-3
    assert np.array_equal(strong_got[:n, 5], strong_ref[:n, 5])              # class indices exact
    agree = (res.masks.data.numpy()[:n].astype(bool) == ref["masks"][:n]).mean()
    assert agree >= 0.995
    # folder-eval style usage (yolo/yolo_folder_eval.py:16-29)
    for r in model.predict(os.path.join(GOLDEN, "bscans"), verbose=False):
        for box in r.boxes:
            x1, y1, x2, y2 = box.xyxy.tolist()[0]
            assert 0 <= x1 <= x2 <= 320 and 0 <= y1 <= y2 <= 320
            float(box.conf), int(box.cls)
        r.names = {0: "FO"}
        assert r.plot().shape == (320, 320, 3)
