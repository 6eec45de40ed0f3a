"""ctypes binding of ``libmi355yolo.so`` (C-ABI declared in ``include/mi355yolo.h``).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be resolved the
import of this module raises, and ``m355_create`` fails on a machine without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355yolo.so")


class M355Error(RuntimeError):
    """Raised when a libmi355yolo call returns a negative status."""


class ModelDesc(C.Structure):
    _fields_ = [("scale", C.c_int), ("nc", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int),
                ("max_batch", C.c_int)]


class ConvInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("cin", C.c_int), ("cout", C.c_int), ("k", C.c_int),
                ("stride", C.c_int), ("has_bn", C.c_int), ("transposed", C.c_int), ("act", C.c_int)]


class OpInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 48), ("layer", C.c_char * 64), ("flops_per_image", C.c_double),
                ("bytes_per_image", C.c_double), ("weight_bytes", C.c_double)]


class AugParams(C.Structure):
    """m355_aug_params (include/mi355yolo.h)."""
    _fields_ = [("src", C.c_int32 * 4), ("xc", C.c_float), ("yc", C.c_float), ("minv", C.c_float * 6),
                ("hgain", C.c_float), ("sgain", C.c_float), ("vgain", C.c_float), ("flip", C.c_int32), ("mosaic", C.c_int32)]


class ConvLaunchArgs(C.Structure):
    """m355_conv_args (include/mi355yolo.h)."""
    _fields_ = [("x", C.c_void_p), ("x_bstride", C.c_int64), ("ldx", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32),
                ("cin", C.c_int32), ("w_packed", C.c_void_p), ("kpad", C.c_int32), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("y_bstride", C.c_int64), ("ldy", C.c_int32), ("ho", C.c_int32), ("wo", C.c_int32),
                ("cout", C.c_int32), ("res", C.c_void_p), ("r_bstride", C.c_int64), ("ldr", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("batch", C.c_int32),
                ("act", C.c_int32), ("out_f32", C.c_int32), ("convt_co", C.c_int32), ("tmode", C.c_int32),
                ("zero_page", C.c_void_p)]


class WgradLaunchArgs(C.Structure):
    """m355_wgrad_args (include/mi355yolo.h)."""
    _fields_ = [("dz", C.c_void_p), ("dz_bstride", C.c_int64), ("lddz", C.c_int32), ("x", C.c_void_p),
                ("x_bstride", C.c_int64), ("ldx", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("cin", C.c_int32),
                ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("batch", C.c_int32), ("dw", C.c_void_p), ("zero_page", C.c_void_p), ("ws", C.c_void_p),
                ("ws_bytes", C.c_int64)]


# symbol -> (restype, argtypes); every entry of include/mi355yolo.h
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SIGNATURES = {
    "m355_version": (C.c_char_p, []),
    "m355_last_error": (C.c_char_p, [_P]),
    "m355_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_P)]),
    "m355_destroy": (None, [_P]),
    "m355_num_convs": (C.c_int, [_P]),
    "m355_get_conv_info": (C.c_int, [_P, C.c_int, C.POINTER(ConvInfo)]),
    "m355_num_anchors": (C.c_int, [_P]),
    "m355_pred_width": (C.c_int, [_P]),
    "m355_proto_hw": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "m355_workspace_bytes": (C.c_size_t, [_P]),
    "m355_flops_per_image": (C.c_double, [_P]),
    "m355_set_conv_weights": (C.c_int, [_P, C.c_int, _P, _P]),
    "m355_forward": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "m355_num_ops": (C.c_int, [_P]),
    "m355_get_op_info": (C.c_int, [_P, C.c_int, C.POINTER(OpInfo)]),
    "m355_set_profiling": (C.c_int, [_P, C.c_int]),
    "m355_collect_op_times": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "m355_get_raw_head": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int)]),
    "m355_copy_raw_head": (C.c_int, [_P, C.c_int, _P, _P]),
    "m355_set_keep_raw": (C.c_int, [_P, C.c_int]),
    "m355_postprocess": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P, _P]),
    "m355_conv2d_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int,
                                  C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "m355_c2f_c32_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "m355_bneck_pair_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, C.c_int, _P]),
    "m355_s2c64_cv1_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "m355_stem_s2c32_cv1_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "m355_proto_phase_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "m355_head_tail_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "m355_tal_assign_launch": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "m355_conv2d_dgrad": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "m355_conv2d_wgrad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "m355_bn_silu_train_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_float, C.c_int, _P, _P, _P, _P, _P]),
    "m355_bn_silu_train_bwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "m355_bn_workspace_floats": (C.c_size_t, [C.c_int]),
    "m355_wgrad_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "m355_grad_sumsq_workspace_floats": (C.c_size_t, []),
    "m355_convt2x2_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "m355_stem_fwd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "m355_sppf_pool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "m355_upsample2x": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "m355_head_decode": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "m355_nms": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P]),
    "m355_conv_launch": (C.c_int, [C.POINTER(ConvLaunchArgs), _P]),
    "m355_wgrad_launch": (C.c_int, [C.POINTER(WgradLaunchArgs), _P]),
    "m355_bn_train_fwd_launch": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_float, C.c_int32, _P, C.c_int32,
                                           _P, C.c_int32, _P, _P, _P, _P, _P, C.c_float, _P]),
    "m355_bn_train_bwd_launch": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32,
                                           _P, C.c_int32, _P, _P, _P]),
    "m355_adamw_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int32, C.c_float, C.c_float, _P]),
    "m355_sgd_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float,
                                C.c_float, C.c_float, _P]),
    "m355_grad_sumsq": (C.c_int, [_P, C.c_int64, _P, _P]),
    "m355_augment": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "m355_msda_forward": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, C.c_int32,
                                    C.c_int32, C.c_int32, _P, _P]),
    "m355_msda_module_forward": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, _P,
                                           C.c_int32, C.c_int32, C.c_float, _P, _P]),
    "m355_dfine_decode": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, C.c_int32, _P]),
    "m355_repack_launch": (C.c_int, [_P, _P, C.c_int32, _P]),
    "m355_sppf_pool_bwd_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int64,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "m355_sppf_pool_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _P]),
    "m355_upsample2x_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, _P]),
    "m355_colsum_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "m355_colsum_launch": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    "m355_upsample2x_bwd_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, _P]),
    "m355_addsilu_fwd_launch": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "m355_addsilu_bwd_launch": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, C.c_int32, _P]),
    "m355_adown_fwd_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _P]),
    "m355_adown_bwd_launch": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, _P]),
    "m355_u8_to_f16x8_launch": (C.c_int, [_P, _P, C.c_int64, _P]),
    "m355_mask_loss_launch": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                         C.c_int32, _P, _P]),
    "m355_box_loss_launch": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, _P, _P, _P]),
    "m355_dfl_decode_launch": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "m355_proto_masks": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, engine=None) -> None:
    if rc < 0:
        msg = lib.m355_last_error(engine)
        raise M355Error(f"libmi355yolo error {rc}: {msg.decode() if msg else '?'}")


def version() -> str:
    return lib.m355_version().decode()
