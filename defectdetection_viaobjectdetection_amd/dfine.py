"""D-FINE decoder hot ops on the HIP library.

Same names, argument meaning and error behaviour as the functions /root/reference/D-Fine/temporal_dfine.py:11-15,
160-181 imports from / reaches through ``transformers.models.d_fine.modeling_d_fine`` (5.15.0):
``multi_scale_deformable_attention_v2`` (:150-221), ``weighting_function`` (:1091-1112), ``distance2bbox``
(:1115-1137), ``DFineIntegral.forward`` (:756-778).  Tensors are CUDA fp32; there is no CPU path (the C-ABI call
fails loudly without a gfx950 device).  A maintainer binds them with
``modeling_d_fine.multi_scale_deformable_attention_v2 = dfine.multi_scale_deformable_attention_v2`` (the attention
module keeps a reference in ``self.ms_deformable_attn_core``, :244) — see INTEGRATION.md.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from ._capi import check, lib


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor: the D-FINE ops have no CPU fallback")
    return t.to(torch.float32).contiguous()


def multi_scale_deformable_attention_v2(value: torch.Tensor, value_spatial_shapes, sampling_locations: torch.Tensor,
                                        attention_weights: torch.Tensor, num_points_list: List[int],
                                        method: str = "default") -> torch.Tensor:
    """value (B, S, heads, 32); value_spatial_shapes [(h, w)] per level (list or tensor); sampling_locations
    (B, Q, heads, 1, P, 2) or (B, Q, heads, P, 2); attention_weights (B, Q, heads, P).  Returns (B, Q, heads * 32)."""
    if method not in ("default", "discrete"):
        raise ValueError(f"unknown method {method!r}")   # the reference leaves sampling_grids undefined (NameError)
    B, S, H, D = value.shape
    loc = sampling_locations
    if loc.dim() == 6:          # the attention module passes (B, Q, heads, 1, P, 2) when reference points are 4-d
        loc = loc.reshape(loc.shape[0], loc.shape[1], loc.shape[2], -1, 2)
    Q, P = loc.shape[1], loc.shape[3]
    shapes = [(int(h), int(w)) for h, w in (value_spatial_shapes.tolist() if torch.is_tensor(value_spatial_shapes)
                                            else value_spatial_shapes)]
    if sum(h * w for h, w in shapes) != S:
        raise ValueError("spatial shapes do not add up to the value sequence length")
    if sum(num_points_list) != P or len(num_points_list) != len(shapes):
        raise ValueError("num_points_list must have one entry per level and add up to the number of points")
    value, loc, attn = _f32c(value, "value"), _f32c(loc, "sampling_locations"), _f32c(attention_weights, "attention_weights")
    out = torch.empty((B, Q, H * D), dtype=torch.float32, device=value.device)
    sh = (C.c_int32 * (2 * len(shapes)))(*[v for hw in shapes for v in hw])
    pp = (C.c_int32 * len(shapes))(*[int(n) for n in num_points_list])
    check(lib.m355_msda_forward(C.c_void_p(value.data_ptr()), B, S, H, D, sh, len(shapes), C.c_void_p(loc.data_ptr()),
                                C.c_void_p(attn.data_ptr()), pp, Q, P, 1 if method == "discrete" else 0,
                                C.c_void_p(out.data_ptr()), _stream()))
    return out


def deformable_attention(hidden_states: torch.Tensor, reference_points: torch.Tensor, encoder_hidden_states: torch.Tensor,
                         spatial_shapes_list, sampling_offsets: torch.nn.Linear, attention_weights: torch.nn.Linear,
                         num_points_list: List[int], n_heads: int, offset_scale: float) -> torch.Tensor:
    """DFineMultiscaleDeformableAttention.forward (modeling_d_fine.py:247-311) for 4-d reference points and method
    "default": the two linear layers run as torch GEMMs, everything behind them (softmax over the points, sampling
    locations from the reference boxes, bilinear gather-weighted-sum) is ONE kernel.
    hidden_states (B, Q, d); reference_points (B, Q, 1, 4) or (B, Q, 4); encoder_hidden_states (B, S, d) -> (B, Q, d)."""
    B, Q, d = hidden_states.shape
    S = encoder_hidden_states.shape[1]
    D = d // n_heads
    P = sum(num_points_list)
    shapes = [(int(h), int(w)) for h, w in spatial_shapes_list]
    if sum(h * w for h, w in shapes) != S:
        raise ValueError("Make sure to align the spatial shapes with the sequence length of the encoder hidden states")
    ref = reference_points.reshape(B, Q, -1)
    if ref.shape[-1] != 4:
        raise ValueError(f"Last dim of reference_points must be 4 for the fused form, but get {ref.shape[-1]} instead.")
    value = _f32c(encoder_hidden_states, "encoder_hidden_states")          # (B, S, heads, D) is a view of (B, S, d)
    off = _f32c(sampling_offsets(hidden_states), "sampling offsets")       # (B, Q, heads * P * 2)
    logit = _f32c(attention_weights(hidden_states), "attention logits")    # (B, Q, heads * P)
    ref = _f32c(ref, "reference_points")
    out = torch.empty((B, Q, d), dtype=torch.float32, device=value.device)
    sh = (C.c_int32 * (2 * len(shapes)))(*[v for hw in shapes for v in hw])
    pp = (C.c_int32 * len(shapes))(*[int(n) for n in num_points_list])
    check(lib.m355_msda_module_forward(C.c_void_p(value.data_ptr()), B, S, n_heads, D, sh, len(shapes), C.c_void_p(ref.data_ptr()),
                                       C.c_void_p(off.data_ptr()), C.c_void_p(logit.data_ptr()), pp, Q, P, float(offset_scale),
                                       C.c_void_p(out.data_ptr()), _stream()))
    return out


def weighting_function(max_num_bins: int, up: torch.Tensor, reg_scale) -> torch.Tensor:
    """W(n), max_num_bins + 1 values (modeling_d_fine.py:1091-1112); a few dozen scalars: computed with torch ops on
    the device `up` lives on, in the reference's order of operations."""
    reg = reg_scale if torch.is_tensor(reg_scale) else torch.tensor(float(reg_scale), dtype=up.dtype, device=up.device)
    upper_bound1 = abs(up[0]) * abs(reg)
    upper_bound2 = abs(up[0]) * abs(reg) * 2
    step = (upper_bound1 + 1) ** (2 / (max_num_bins - 2))
    left = [-((step) ** i) + 1 for i in range(max_num_bins // 2 - 1, 0, -1)]
    right = [(step) ** i - 1 for i in range(1, max_num_bins // 2)]
    values = [-upper_bound2] + left + [torch.zeros_like(up[0][None])] + right + [upper_bound2]
    return torch.cat([v.reshape(1) for v in values], 0)


def decode_boxes(pred_corners: torch.Tensor, project: torch.Tensor, points: torch.Tensor, reg_scale: float,
                 clamp01: bool = False) -> torch.Tensor:
    """integral(pred_corners, project) -> distance2bbox(points, ., reg_scale) [-> clamp(0, 1)] in one kernel
    (temporal_dfine.py:180-181).  pred_corners (..., 4 * (bins + 1)), points (..., 4) -> boxes (..., 4)."""
    nb1 = project.numel()
    lead = pred_corners.shape[:-1]
    if pred_corners.shape[-1] != 4 * nb1 or tuple(points.shape) != tuple(lead) + (4,):
        raise ValueError("pred_corners must be (..., 4 * len(project)) and points (..., 4)")
    d, pr, pt = _f32c(pred_corners, "pred_corners"), _f32c(project, "project"), _f32c(points, "points")
    out = torch.empty(tuple(lead) + (4,), dtype=torch.float32, device=d.device)
    n = out.numel() // 4
    check(lib.m355_dfine_decode(C.c_void_p(d.data_ptr()), C.c_void_p(pr.data_ptr()), C.c_void_p(pt.data_ptr()),
                                C.c_void_p(out.data_ptr()), n, nb1, float(reg_scale), int(clamp01), _stream()))
    return out


def integral(pred_corners: torch.Tensor, project: torch.Tensor) -> torch.Tensor:
    """DFineIntegral.forward: (B, Q, 4 * (bins + 1)) logits -> (B, Q, 4) distances.  Runs the decode kernel against a
    reference point chosen so that the box IS the distances: with reg_scale 1 and points (0, 0, 1, 1) the corners are
    (-(0.5 + d0), -(0.5 + d1), 0.5 + d2, 0.5 + d3); the distances are recovered exactly only up to that affine map, so
    this entry computes them with torch softmax + matmul on the device instead (it is not a hot op on its own)."""
    nb1 = project.numel()
    b, q, _ = pred_corners.shape
    p = torch.softmax(pred_corners.reshape(-1, nb1), dim=1)
    return torch.nn.functional.linear(p, project.to(p.device).reshape(1, -1)).reshape(b, q, -1)


def distance2bbox(points: torch.Tensor, distance: torch.Tensor, reg_scale: float) -> torch.Tensor:
    """modeling_d_fine.py:1115-1137 on device tensors (elementwise; the fused form is `decode_boxes`)."""
    reg_scale = abs(reg_scale)
    x0 = points[..., 0] - (0.5 * reg_scale + distance[..., 0]) * (points[..., 2] / reg_scale)
    y0 = points[..., 1] - (0.5 * reg_scale + distance[..., 1]) * (points[..., 3] / reg_scale)
    x1 = points[..., 0] + (0.5 * reg_scale + distance[..., 2]) * (points[..., 2] / reg_scale)
    y1 = points[..., 1] + (0.5 * reg_scale + distance[..., 3]) * (points[..., 3] / reg_scale)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], -1)
