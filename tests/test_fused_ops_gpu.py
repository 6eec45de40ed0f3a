"""The fused launches of round 3 -- conv3x3_s2c64 + 1x1, the stem + model.1 + model.2.cv1 launch (lockstep and two-team forms), the
composed Proto launch, a head level's output convs + decode -- each through its per-op C-ABI entry against a plain PyTorch fp32
reference with the engine's rounding points (fp16 inputs / weights, fp32 sums, fp16 where a tensor is stored or handed to the next
MFMA).  Round 3 checked these kernels against the older HIP kernels only (and, inside the whole network, against the oracle).
Upstream modules restated: Conv (Conv2d + folded BN + SiLU), ConvTranspose2d, Proto, Detect's DFL + dist2bbox."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

r16 = lambda t: t.half().float()                                       # noqa: E731
h_ = lambda a: a.ctypes.data_as(C.c_void_p)                            # noqa: E731
f32 = lambda t: t.numpy().astype(np.float32).copy()                    # noqa: E731


def _rel(got, want):
    return float((got - want).norm() / want.norm())


@pytest.mark.parametrize("B,H,W", [(6, 160, 160), (2, 32, 48), (1, 16, 16)])   # 6 x 160 x 160 = the shape that dispatches it inside the engine at 640 x 640
def test_s2c64_cv1_against_torch(cuda_device, B, H, W):
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(H + B)
    x = (torch.randn((B, 64, H, W), generator=g) * 0.8).half()
    w3 = r16(torch.randn((128, 64, 3, 3), generator=g) * (2.0 / (9 * 64)) ** 0.5)
    w1 = r16(torch.randn((128, 128, 1, 1), generator=g) * (2.0 / 128) ** 0.5)
    b3, b1 = torch.randn(128, generator=g) * 0.3, torch.randn(128, generator=g) * 0.3
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda_device)
    yd = torch.full((B, H // 2, W // 2, 128), float("nan"), dtype=torch.float16, device=cuda_device)
    a = [f32(t) for t in (w3, b3, w1, b1)]
    _capi.check(_capi.lib.m355_s2c64_cv1_fwd(C.c_void_p(xd.data_ptr()), B, H, W, h_(a[0]), h_(a[1]), h_(a[2]), h_(a[3]),
                                             C.c_void_p(yd.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    t = r16(F.silu(F.conv2d(x.float(), w3, b3, stride=2, padding=1)))
    want = F.silu(F.conv2d(t, w1, b1))
    got = yd.float().cpu().permute(0, 3, 1, 2)
    rel = _rel(got, want)
    print(f"s2c64+1x1 B={B} {H}x{W}: rel-L2 {rel:.2e}")
    assert torch.isfinite(got).all() and rel <= 1e-3


@pytest.mark.parametrize("two_team", [1, 0])
@pytest.mark.parametrize("B,H,W", [(3, 320, 320), (2, 64, 128)])
def test_stem_launch_against_torch(cuda_device, B, H, W, two_team):
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(H + two_team)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    w0 = r16(torch.randn((32, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5 * 2.0)       # (the engine stores the stem weights unscaled in fp16)
    w1 = r16(torch.randn((64, 32, 3, 3), generator=g) * (2.0 / (9 * 32)) ** 0.5)
    w2 = r16(torch.randn((64, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5)
    b0, b1, b2 = (torch.randn(n, generator=g) * 0.3 for n in (32, 64, 64))
    xd = img.to(cuda_device)
    yd = torch.full((B, H // 4, W // 4, 64), float("nan"), dtype=torch.float16, device=cuda_device)
    a = [f32(t) for t in (w0, b0, w1, b1, w2, b2)]
    _capi.check(_capi.lib.m355_stem_s2c32_cv1_fwd(C.c_void_p(xd.data_ptr()), B, H, W, h_(a[0]), h_(a[1]), h_(a[2]), h_(a[3]), h_(a[4]), h_(a[5]),
                                                  C.c_void_p(yd.data_ptr()), two_team, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    x = img.permute(0, 3, 1, 2).float()
    t0 = r16(F.silu(F.conv2d(x, w0, None, stride=2, padding=1) * (1.0 / 255.0) + b0.view(1, -1, 1, 1)))
    t1 = r16(F.silu(F.conv2d(t0, w1, b1, stride=2, padding=1)))
    want = F.silu(F.conv2d(t1, w2, b2))
    got = yd.float().cpu().permute(0, 3, 1, 2)
    rel = _rel(got, want)
    print(f"stem launch (two_team={two_team}) B={B} {H}x{W}: rel-L2 {rel:.2e}")
    assert torch.isfinite(got).all() and rel <= 1e-3


@pytest.mark.parametrize("B,H,W", [(2, 80, 80), (3, 16, 32)])
def test_proto_phase_launch_against_torch(cuda_device, B, H, W):
    """ConvTranspose2d(128, 128, 2, 2, bias) -> Conv3x3 + SiLU -> Conv1x1 (128 -> 32) + SiLU.  The kernel runs the first two as four
    2x2 phase convs with weights composed on the host in fp64 and rounded to fp16 ONCE (the reference uses the un-composed fp32
    weights: the bound allows for that one rounding, ~3e-4), the 128-channel tile after the SiLU is fp16 in LDS."""
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(W)
    x = (torch.randn((B, 128, H, W), generator=g) * 0.8).half()
    wt = torch.randn((128, 128, 2, 2), generator=g) * (1.0 / 128) ** 0.5
    w3 = torch.randn((128, 128, 3, 3), generator=g) * (2.0 / (9 * 128)) ** 0.5
    wc = r16(torch.randn((32, 128, 1, 1), generator=g) * (2.0 / 128) ** 0.5)
    bt, b3, bc = (torch.randn(n, generator=g) * 0.3 for n in (128, 128, 32))
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda_device)
    yd = torch.full((B, 2 * H, 2 * W, 32), float("nan"), dtype=torch.float16, device=cuda_device)
    a = [f32(t) for t in (wt, bt, w3, b3, wc, bc)]
    _capi.check(_capi.lib.m355_proto_phase_fwd(C.c_void_p(xd.data_ptr()), B, H, W, h_(a[0]), h_(a[1]), h_(a[2]), h_(a[3]), h_(a[4]), h_(a[5]),
                                               C.c_void_p(yd.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    up = F.conv_transpose2d(x.double(), wt.double(), bt.double(), stride=2)
    z = r16(F.silu(F.conv2d(up, w3.double(), b3.double(), padding=1)).float())
    want = F.silu(F.conv2d(z, wc, bc))
    got = yd.float().cpu().permute(0, 3, 1, 2)
    rel = _rel(got, want)
    print(f"proto launch B={B} {H}x{W}: rel-L2 {rel:.2e}")
    assert torch.isfinite(got).all() and rel <= 1.5e-3


@pytest.mark.parametrize("B,H,W,nc,stride", [(3, 40, 40, 1, 16.0), (2, 80, 80, 3, 8.0), (5, 10, 10, 20, 32.0), (2, 19, 20, 1, 32.0)])
def test_head_tail_against_torch(cuda_device, B, H, W, nc, stride):
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(H * 7 + nc)
    x = (torch.randn((B, H, W, 224), generator=g) * 0.8).half()
    w2 = r16(torch.randn((64, 64, 1, 1), generator=g) * 0.2)
    w3 = r16(torch.randn((nc, 128, 1, 1), generator=g) * 0.1)
    w4 = r16(torch.randn((32, 32, 1, 1), generator=g) * 0.2)
    b2, b3, b4 = torch.randn(64, generator=g) * 0.5 + 1.0, torch.randn(nc, generator=g) - 2.0, torch.randn(32, generator=g) * 0.3
    A, off = H * W + 37, 21                                                # the level sits inside a longer anchor axis
    preds = torch.full((B, A, 4 + nc + 32), -7.0, dtype=torch.float32, device=cuda_device)
    xd = x.to(cuda_device)
    a = [f32(t) for t in (w2, b2, w3, b3, w4, b4)]
    _capi.check(_capi.lib.m355_head_tail_fwd(C.c_void_p(xd.data_ptr()), B, H, W, nc, C.c_float(stride), h_(a[0]), h_(a[1]), h_(a[2]), h_(a[3]),
                                             h_(a[4]), h_(a[5]), C.c_void_p(preds.data_ptr()), A, off,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    xc = x.float().permute(0, 3, 1, 2)
    box = F.conv2d(xc[:, :64], w2, b2).view(B, 4, 16, H * W)
    cls = F.conv2d(xc[:, 64:192], w3, b3).view(B, nc, H * W)
    coef = F.conv2d(xc[:, 192:], w4, b4).view(B, 32, H * W)
    dist = (box.softmax(2) * torch.arange(16.0).view(1, 1, 16, 1)).sum(2)   # (B, 4, HW): l, t, r, b
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    anc = torch.stack((xs.reshape(-1), ys.reshape(-1)), 0).unsqueeze(0)      # (1, 2, HW)
    x1y1, x2y2 = anc - dist[:, :2], anc + dist[:, 2:]
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * stride
    want = torch.cat((dbox, cls.sigmoid(), coef), 1).permute(0, 2, 1)        # (B, HW, 4 + nc + 32)
    got = preds.cpu()
    assert bool((got[:, :off] == -7.0).all()) and bool((got[:, off + H * W:] == -7.0).all())   # rows of other levels untouched
    lv = got[:, off:off + H * W]
    assert torch.isfinite(lv).all()
    e_box = float((lv[..., :4] - want[..., :4]).abs().max())
    e_cls = float((lv[..., 4:4 + nc] - want[..., 4:4 + nc]).abs().max())
    e_coef = _rel(lv[..., 4 + nc:], want[..., 4 + nc:])
    print(f"head_tail B={B} {H}x{W} nc={nc}: box max |d| {e_box:.2e} px, score max |d| {e_cls:.2e}, coef rel-L2 {e_coef:.2e}")
    assert e_box <= 2e-3 * stride and e_cls <= 1e-4 and e_coef <= 1e-4
