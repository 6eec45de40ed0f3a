"""The CPU oracle in the ENGINE'S NUMBER FORMAT  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``tools/`` diagnostics and ``__graft_entry__.smoke()`` may import this module.

``north_star`` / SURVEY 8a row A4 prescribe the arithmetic of the MI355X path: fp16 NHWC storage of weights and
activations, fp32 accumulation.  ``yolov8_seg_oracle.SegmentationModel`` is the fp32 restatement of the algorithm
(what upstream's CPU path computes); this module takes such a model and applies the storage format AT THE POINTS
WHERE THE ENGINE STORES -- nothing else changes, every sum is still PyTorch-CPU fp32:

  * Conv+BN: BN folded in fp64, rounded to fp32, then to fp16 (``spec.fold_bn`` + ``engine.hip:pack_conv_rows``);
    bias stays fp32; ``y = fp16(silu(conv(x, w16) + b))``;
  * Bottleneck with shortcut: ``fp16(x + silu(...))`` -- ONE rounding after the residual add (conv epilogue);
  * stem: integer pixels x fp16 weights, ``* (1/255) + bias`` in fp32 (``misc_kernels.hip:stem_rows_kernel``);
  * head output convs (``cv{2,3,4}.{l}.2``): fp16 weights, fp32 bias, fp32 output (raw head map);
  * Proto: ConvTranspose2d(2x2, s2, bias) -> Conv3x3 composed on the host in fp64 into four 2x2 phase
    convolutions over the low-resolution tensor with a 3x3 border-class bias table, composed weights rounded to
    fp16 (``engine.hip:m355_set_conv_weights``, composed branch); no rounding at the ConvTranspose output;
  * max-pool, nearest upsample, concat: exact.

Two uses (tests/test_engine_gpu.py, tests/test_keepset_gpu.py):
  1. the HIP path against THIS model differs only by fp32 summation order, the SiLU approximation and the rare 1-ulp
     fp16 flips those cause -- a bound ~20x tighter than against the fp32 oracle, so it catches real kernel bugs;
  2. THIS model against the fp32 oracle is what the number format itself costs (tools/fp16_floor.py): the HIP
     path's distance from the fp32 oracle is asserted against that measured floor, not against a guessed constant.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

import yolov8_seg_oracle as orc

BN_EPS = 1e-3


def _fold(conv_mod: "orc.Conv"):
    bn = conv_mod.bn
    scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    w = (conv_mod.conv.weight.double() * scale.view(-1, 1, 1, 1)).float()
    b = (bn.bias.double() - bn.running_mean.double() * scale).float()
    return w, b


def _h(x: torch.Tensor) -> torch.Tensor:
    return x.half().float()


def _compose_proto(wt, bt, w3, b3):
    """Weff[q] (n, n, 2, 2) per phase q = py * 2 + px, and the (3, 3, n) bias table -- engine.hip composed branch."""
    n = w3.shape[0]
    wt, w3, bt, b3 = wt.double(), w3.double(), bt.double(), b3.double()
    weff = torch.zeros(4, n, n, 2, 2, dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            for kh in range(3):
                ty = py + kh - 1
                ry, dy = (-1 if ty < 0 else ty // 2), ty & 1
                a = ry + 1 - py
                for kw in range(3):
                    tx = px + kw - 1
                    rx, dx = (-1 if tx < 0 else tx // 2), tx & 1
                    b = rx + 1 - px
                    # sum_c W3[co, c, kh, kw] * Wt[ci, c, dy, dx]
                    weff[py * 2 + px, :, :, a, b] += w3[:, :, kh, kw] @ wt[:, :, dy, dx].t()
    btab = torch.zeros(3, 3, n, dtype=torch.float64)
    for ry in range(3):
        for rx in range(3):
            s = b3.clone()
            for kh in range(3):
                if (ry == 0 and kh == 0) or (ry == 2 and kh == 2):
                    continue
                for kw in range(3):
                    if (rx == 0 and kw == 0) or (rx == 2 and kw == 2):
                        continue
                    s += w3[:, :, kh, kw] @ bt
            btab[ry, rx] = s
    return _h(weff.float()), btab.float()


def to_engine_format(model: "orc.SegmentationModel", composed_proto: bool = True) -> "orc.SegmentationModel":
    """Patches ``model`` (eval mode, weights loaded) in place and returns it."""
    model.eval()
    stem = model.model[0]
    for mod in model.modules():
        if isinstance(mod, orc.Conv) and mod is not stem:
            w, b = _fold(mod)
            w = _h(w)
            mod.forward = (lambda x, w=w, b=b, c=mod.conv: _h(F.silu(F.conv2d(x, w, b, c.stride, c.padding))))
    w0, b0 = _fold(stem)
    w0 = _h(w0)
    inv255 = torch.tensor(1.0 / 255.0, dtype=torch.float32)
    stem.forward = (lambda x, w=w0, b=b0, c=stem.conv:
                    _h(F.silu(F.conv2d(torch.round(x * 255.0), w, None, c.stride, c.padding) * inv255 + b.view(1, -1, 1, 1))))
    for mod in model.modules():
        if isinstance(mod, orc.Bottleneck) and mod.add:
            w, b = _fold(mod.cv2)
            w = _h(w)
            mod.forward = (lambda x, mod=mod, w=w, b=b:
                           _h(x + F.silu(F.conv2d(mod.cv1(x), w, b, mod.cv2.conv.stride, mod.cv2.conv.padding))))
    seg = model.model[22]
    for branch in (seg.cv2, seg.cv3, seg.cv4):
        for seq in branch:
            last = seq[-1]
            w = _h(last.weight.data.float())
            seq[-1].forward = (lambda x, w=w, b=last.bias.data.float(): F.conv2d(x, w, b))
    proto = seg.proto
    if composed_proto:
        w3, b3 = _fold(proto.cv2)
        weff, btab = _compose_proto(proto.upsample.weight.data.float(), proto.upsample.bias.data.float(), w3, b3)

        def proto_fwd(x, proto=proto, weff=weff, btab=btab):
            z = proto.cv1(x)
            bsz, n, hh, ww = z.shape
            out = torch.empty(bsz, n, 2 * hh, 2 * ww)
            cls_y = torch.ones(2 * hh, dtype=torch.long)
            cls_y[0], cls_y[-1] = 0, 2
            cls_x = torch.ones(2 * ww, dtype=torch.long)
            cls_x[0], cls_x[-1] = 0, 2
            bias_map = btab[cls_y][:, cls_x].permute(2, 0, 1)           # (n, 2H, 2W)
            for py in range(2):
                for px in range(2):
                    zp = F.pad(z, (1 - px, px, 1 - py, py))
                    out[:, :, py::2, px::2] = F.conv2d(zp, weff[py * 2 + px])
            return proto.cv3(_h(F.silu(out + bias_map.unsqueeze(0))))
        proto.forward = proto_fwd
    else:
        wt = _h(proto.upsample.weight.data.float())
        proto.upsample.forward = (lambda x, wt=wt, bt=proto.upsample.bias.data.float(): _h(F.conv_transpose2d(x, wt, bt, 2)))
    return model
