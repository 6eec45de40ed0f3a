"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mi355yolo.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355yolo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m355_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("m355_create", "m355_destroy", "m355_forward", "m355_postprocess", "m355_set_conv_weights",
                 "m355_last_error", "m355_conv2d_fwd", "m355_nms", "m355_proto_masks", "m355_head_decode"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib_path = os.path.join(ROOT, "defectdetection_viaobjectdetection_amd", "lib", "libmi355yolo.so")
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_header():
    from defectdetection_viaobjectdetection_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_symbols()
    assert "gfx950" in _capi.version()


def test_no_device_is_a_loud_error():
    """Without a GPU m355_create must fail with M355_ERR_NO_DEVICE (-2): there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    from defectdetection_viaobjectdetection_amd import _capi
    desc = _capi.ModelDesc(ord("n"), 1, 64, 64, 1)
    h = ctypes.c_void_p()
    rc = _capi.lib.m355_create(ctypes.byref(desc), ctypes.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in _capi.lib.m355_last_error(None)


def test_forced_tile_ids_are_validated_before_any_device_work():
    """m355_conv2d_fwd(force_tile=...) is a debug entry: a tile id no launcher implements (the experimental 256-row channel
    tiles that once faulted the GPU) must come back as M355_ERR_INVALID (-1), decided on the host
    before any allocation or launch -- so the check runs here, without a GPU."""
    import numpy as np
    from defectdetection_viaobjectdetection_amd import _capi
    w = np.zeros((128, 64, 1, 1), np.float32)
    b = np.zeros(128, np.float32)
    fake_dev = ctypes.c_void_p(0x1000)   # never dereferenced: the rejection precedes every HIP call
    for bad in (4, 6, 7, 21, 22, 23, 24, 33, 34, 200, 255, 0x100 | 7):   # (21-24: the lean template removed in round 3; 30-32 exist; 33 = the row-slab kernel: 3x3 only, this call is 1x1)
        rc = _capi.lib.m355_conv2d_fwd(fake_dev, 1, 8, 8, 64, w.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                       128, 1, 1, 1, None, fake_dev, 0, bad, None)
        assert rc == -1, (bad, rc)
        assert b"forced tile" in _capi.lib.m355_last_error(None)
