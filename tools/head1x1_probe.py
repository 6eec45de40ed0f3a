"""The head's output 1x1 (224 -> 64 + nc + 32 channels, fp32 rows of the raw map) in isolation at batch 32, against variants that
change only how its result is stored: ragged fp32 rows (ldy = 97, the network's form), fp32 rows padded to 128 floats, fp16 rows.
Event-timed, 30 launches each.  python tools/head1x1_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd._capi import ConvLaunchArgs, check, lib  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
zero_page = torch.zeros(256, dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(hw, cin, cout, out_f32, ldy, label):
    x = torch.randn(B, hw, hw, cin, device=dev).half()
    kpad = (cin + 63) // 64 * 64
    w = (torch.randn((cout + 127) // 128 * 128, kpad, device=dev) * 0.05).half()
    bias = torch.zeros(cout + 256, device=dev)
    y = torch.empty(B * hw * hw * ldy + 64, dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    a = ConvLaunchArgs()
    a.x, a.x_bstride, a.ldx, a.hi, a.wi, a.cin = x.data_ptr(), hw * hw * cin, cin, hw, hw, cin
    a.w_packed, a.kpad, a.bias = w.data_ptr(), kpad, bias.data_ptr()
    a.y, a.y_bstride, a.ldy, a.ho, a.wo, a.cout = y.data_ptr(), hw * hw * ldy, ldy, hw, hw, cout
    a.res, a.r_bstride, a.ldr = 0, 0, 0
    a.ksize, a.stride, a.pad, a.batch = 1, 1, 0, B
    a.act, a.out_f32, a.convt_co, a.tmode = 0, out_f32, 0, 0
    a.zero_page = zero_page.data_ptr()
    for _ in range(3):
        check(lib.m355_conv_launch(C.byref(a), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        check(lib.m355_conv_launch(C.byref(a), st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = B * hw * hw * (cin * 2 + cout * (4 if out_f32 else 2)) / 1e6
    print(f"{label:44s} {hw}^2 {cin}->{cout}: {us:7.1f} us  {mb:6.1f} MB  {mb / us:5.2f} TB/s")


for hw in (80, 40):
    run(hw, 224, 97, 1, 97, "fp32 rows, ldy 97 (network)")
    run(hw, 224, 97, 1, 128, "fp32 rows, ldy 128")
    run(hw, 224, 104, 0, 104, "fp16 rows, ldy 104")
    run(hw, 224, 128, 0, 128, "fp16 rows, 128 channels")
    run(hw, 256, 128, 0, 128, "fp16, 256 -> 128")
