"""Known answers pinning the YOLOv9c-seg restatement (oracle/yolov9c_seg_oracle.py, spec.conv_specs_v9c) -- row N4.  CPU.
The reference names this architecture (/root/reference/BscanBased/yolo_seg_train.py:7-8, yolo8_seg_predict.py:4) and
holds nothing that pins its outputs; the published model summaries pin every layer shape, closed forms pin the new ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yolov9c_seg_oracle as o9
from defectdetection_viaobjectdetection_amd.spec import (conv_specs, count_parameters, fold_bn, init_state_dict, state_dict_keys,
                                                          synthetic_state_dict)


def test_parameter_counts_equal_the_published_summaries():
    assert sum(p.numel() for p in o9.SegmentationModelV9c(80).parameters()) == 27_897_120     # yolov9c-seg summary
    assert sum(p.numel() for p in o9.DetectionModelV9cCount(80).parameters()) == 25_590_912   # yolov9c summary (Detect head)
    for nc, want in ((80, 27_897_120), (1, 27_836_211)):
        sd = init_state_dict("9c", nc, seed=0)
        assert count_parameters(sd) == want
        model = o9.SegmentationModelV9c(nc)
        assert set(model.state_dict()) == set(state_dict_keys("9c", nc))                      # upstream key names on both sides
        model.load_state_dict(sd, strict=True)


def test_output_shapes_and_spec_list():
    m = o9.SegmentationModelV9c(1).eval()
    with torch.no_grad():
        y, p = m(torch.rand(1, 3, 128, 192))
    assert y.shape == (1, 37, 16 * 24 + 8 * 12 + 4 * 6) and p.shape == (1, 32, 32, 48)
    specs = conv_specs("9c", 1)
    assert len(specs) == 157 and sum(s.rep for s in specs) == 16                              # 8 GELAN blocks x 2 RepConvN
    assert [s.name for s in specs[:4]] == ["model.0", "model.1", "model.2.cv1", "model.2.cv2.0.cv1"]


def test_repconvn_is_one_3x3_conv_after_branch_fusion():
    """fold_bn merges the 3x3 and 1x1 branches (each with its own BN) into the single conv the engine runs."""
    sd = synthetic_state_dict("9c", 1, seed=1)
    spec = next(s for s in conv_specs("9c", 1) if s.rep)
    m = o9.RepConvN(spec.cin, spec.cout).eval()
    m.load_state_dict({k[len(spec.name) + 1:]: v for k, v in sd.items() if k.startswith(spec.name + ".")})
    x = torch.randn(2, spec.cin, 9, 11)
    w, b = fold_bn(sd, spec)
    assert w.shape == (spec.cout, spec.cin, 3, 3)
    with torch.no_grad():
        ref = m(x)
    got = F.silu(F.conv2d(x, w, b, 1, 1))
    assert float((got - ref).abs().max()) <= 2e-5


def test_adown_pooling_closed_form():
    ad = o9.ADown(8, 8).eval()
    x = torch.arange(2 * 8 * 6 * 6, dtype=torch.float32).view(2, 8, 6, 6)
    a = F.avg_pool2d(x, 2, 1, 0)
    assert a.shape == (2, 8, 5, 5) and float(a[0, 0, 0, 0]) == (0 + 1 + 6 + 7) / 4
    mp = F.max_pool2d(a[:, 4:], 3, 2, 1)
    assert mp.shape == (2, 4, 3, 3)
    assert float(mp[0, 0, 0, 0]) == float(a[0, 4, :2, :2].max()) and float(mp[0, 0, 2, 2]) == float(a[0, 4, 3:5, 3:5].max())
    with torch.no_grad():
        y = ad(x)
    assert y.shape == (2, 8, 3, 3)


def test_sppelan_serial_pools_equal_growing_windows():
    sp = o9.SPPELAN(16, 16, 8).eval()
    x = torch.randn(1, 16, 12, 12)
    with torch.no_grad():
        a = sp.cv1(x)
        y = [a]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        assert torch.equal(y[2], F.max_pool2d(a, 9, 1, 4)) and torch.equal(y[3], F.max_pool2d(a, 13, 1, 6))
        assert torch.equal(sp(x), sp.cv5(torch.cat(y, 1)))
