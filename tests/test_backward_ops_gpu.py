"""Backward building blocks (SURVEY A13 backward) vs PyTorch autograd on the CPU (fp32 on fp16-rounded operands)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def nhwc16(x_nchw, dev):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(dev)


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


DGRAD_CASES = [
    # B, H, W, cin, cout, k, stride
    (2, 16, 16, 64, 64, 3, 1),       # halo path
    (2, 20, 24, 32, 64, 3, 1),       # im2col path (Cin of the dgrad GEMM = cout = 64, output channels 32)
    (2, 16, 16, 128, 64, 1, 1),
    (2, 32, 32, 64, 128, 3, 2),      # stride 2 on even maps: four phase convs over dY, pixel-shuffle fast stores (Wo % 16 == 0)
    (1, 34, 22, 32, 64, 3, 2),       # the same with four phases per channel tile (cin 32) and odd dY sizes (17 x 11): generic epilogue
    (3, 40, 40, 128, 256, 3, 2),
    (2, 64, 64, 32, 64, 3, 2),       # cin 32 / cout 64 with dY rows in 32-pixel chunks: the wave-private chunk stream (conv_dgrad_s2c32.hip)
    (3, 128, 192, 32, 64, 3, 2),     # ... three chunks per row, last rows / columns of the image (zero neighbours)
    (9, 320, 320, 32, 64, 3, 2),     # ... more chunks (7 200) than waves in flight (2 048): the register prefetch of a wave's next chunk
    (2, 33, 21, 64, 128, 3, 2),      # odd input sizes: the transposed-stride gather (all nine taps masked per output parity)
    (1, 48, 80, 256, 512, 3, 2),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv2d_dgrad(case, cuda_device):
    from defectdetection_viaobjectdetection_amd import _capi
    B, H, W, cin, cout, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    x = torch.randn(B, cin, H, W, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    d_dy = nhwc16(dy, cuda_device)
    d_dx = torch.full((B, H, W, cin), float("nan"), dtype=torch.float16, device=cuda_device)
    _capi.check(_capi.lib.m355_conv2d_dgrad(_p(d_dy), B, H, W, cin, _p(w.contiguous()), cout, k, s, _p(d_dx),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    got = d_dx.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert rel_l2(got, x.grad) <= 1e-3


def test_stride2_dgrad_phase_form_equals_the_gather_form(cuda_device, monkeypatch):
    """A/B of the two stride-2 input-gradient forms on the same operands: the same nine products per output in fp32, so the results
    agree to fp16 rounding of sums taken in a different order."""
    from defectdetection_viaobjectdetection_amd import _capi
    B, H, W, cin, cout = 2, 40, 48, 128, 256
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).half().float().contiguous()
    d_dy = nhwc16(torch.randn(B, cout, H // 2, W // 2, generator=g), cuda_device)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("M355_NO_DGRAD_PHASES", "1")
        d_dx = torch.full((B, H, W, cin), float("nan"), dtype=torch.float16, device=cuda_device)
        _capi.check(_capi.lib.m355_conv2d_dgrad(_p(d_dy), B, H, W, cin, _p(w), cout, 3, 2, _p(d_dx),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        outs.append(d_dx.float().cpu())
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[0], outs[1]) <= 5e-4


@pytest.mark.parametrize("shape,act", [((4, 20, 20, 64), 1), ((2, 40, 24, 128), 1), ((3, 16, 16, 48), 1), ((2, 8, 8, 512), 0)])
def test_bn_silu_train_fwd_bwd(shape, act, cuda_device):
    """Train-mode BatchNorm(+SiLU) forward and backward vs torch (batch statistics, eps 1e-3)."""
    from defectdetection_viaobjectdetection_amd import _capi
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(Cc + H)
    z = (torch.randn(B, Cc, H, W, generator=g) * 1.7 + 0.3).half().float().requires_grad_(True)
    gamma = (0.8 + 0.4 * torch.rand(Cc, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.rand(Cc, generator=g) - 0.1).requires_grad_(True)
    u = F.batch_norm(z, None, None, gamma, beta, True, 0.0, 1e-3)
    y = F.silu(u) if act else u
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    dev = cuda_device
    d_z, d_dy = nhwc16(z.detach(), dev), nhwc16(dy, dev)
    d_g, d_b = gamma.detach().to(dev), beta.detach().to(dev)
    d_y = torch.empty((B, H, W, Cc), dtype=torch.float16, device=dev)
    d_mean = torch.empty(Cc, device=dev)
    d_is = torch.empty(Cc, device=dev)
    d_ws = torch.zeros(int(_capi.lib.m355_bn_workspace_floats(Cc)), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _capi.check(_capi.lib.m355_bn_silu_train_fwd(_p(d_z), B, H, W, Cc, _p(d_g), _p(d_b), 1e-3, act, _p(d_y), _p(d_mean),
                                                 _p(d_is), _p(d_ws), st))
    d_dz = torch.empty_like(d_y)
    d_gb = torch.empty(2 * Cc, device=dev)
    _capi.check(_capi.lib.m355_bn_silu_train_bwd(_p(d_z), _p(d_dy), B, H, W, Cc, _p(d_mean), _p(d_is), _p(d_g), _p(d_b), act,
                                                 _p(d_dz), _p(d_gb), _p(d_ws), st))
    torch.cuda.synchronize()
    # ordered cross-block reductions (no float atomics): a second run gives the same bits
    y1, dz1, gb1, m1 = d_y.clone(), d_dz.clone(), d_gb.clone(), d_mean.clone()
    _capi.check(_capi.lib.m355_bn_silu_train_fwd(_p(d_z), B, H, W, Cc, _p(d_g), _p(d_b), 1e-3, act, _p(d_y), _p(d_mean),
                                                 _p(d_is), _p(d_ws), st))
    _capi.check(_capi.lib.m355_bn_silu_train_bwd(_p(d_z), _p(d_dy), B, H, W, Cc, _p(d_mean), _p(d_is), _p(d_g), _p(d_b), act,
                                                 _p(d_dz), _p(d_gb), _p(d_ws), st))
    torch.cuda.synchronize()
    assert torch.equal(y1, d_y) and torch.equal(dz1, d_dz) and torch.equal(gb1, d_gb) and torch.equal(m1, d_mean)
    zd = z.detach()
    assert torch.allclose(d_mean.cpu(), zd.mean((0, 2, 3)), atol=1e-4)
    assert torch.allclose(d_is.cpu(), 1 / torch.sqrt(zd.var((0, 2, 3), unbiased=False) + 1e-3), rtol=1e-4)
    assert rel_l2(d_y.float().cpu().permute(0, 3, 1, 2), y.detach()) <= 1e-3
    assert rel_l2(d_dz.float().cpu().permute(0, 3, 1, 2), z.grad) <= 2e-3
    assert rel_l2(d_gb[:Cc].cpu(), beta.grad) <= 1e-3
    assert rel_l2(d_gb[Cc:].cpu(), gamma.grad) <= 1e-3


WGRAD_CASES = [
    # B, H, W, cin, cout, k, stride
    (2, 16, 16, 64, 64, 3, 1),
    (2, 20, 24, 32, 64, 3, 1),        # Cin = 32: an n-tile spans four taps
    (3, 17, 13, 128, 128, 3, 1),      # ragged pixel count (M not a multiple of 64)
    (2, 32, 32, 64, 128, 3, 2),
    (2, 16, 16, 256, 128, 1, 1),
    (2, 16, 16, 96, 48, 1, 1),        # ragged channel tiles on both sides
    (8, 40, 40, 128, 224, 3, 1),      # many split-K blocks; patch kernel, 64 x 64 channel tiles, ragged tile columns (40 = 2.5 x 16)
    # 3x3 / s1 maps at least two tile columns wide take the patch kernel (conv_wgrad3.hip): every wave layout
    (2, 16, 32, 64, 64, 3, 1),        # 2 x 2 channel blocks, one per wave
    (2, 24, 40, 32, 32, 3, 1),        # 1 x 1: the four waves split the tile rows
    (1, 16, 32, 128, 32, 3, 1),       # 1 x 2
    (1, 19, 35, 32, 64, 3, 1),        # 2 x 1, ragged rows and columns
    (2, 16, 48, 48, 96, 3, 1),        # m-scale widths: half-empty channel tiles on both sides
    # the stem (3x3 / s2 on the 8-channel padded input rows, Cout <= 64, output rows in 64-pixel chunks): wgrad_stem_kernel
    (2, 128, 128, 8, 32, 3, 2),       # s scale: one 32-channel block, first / last rows and columns zero-padded
    (1, 64, 256, 8, 48, 3, 2),        # m scale: two channel blocks, the second half empty
    (3, 32, 128, 8, 16, 3, 2),        # n scale
    (5, 16, 384, 8, 64, 3, 2),        # two full channel blocks
    (4, 320, 512, 8, 32, 3, 2),       # 2 560 chunks on 1 536 waves: the register prefetch of a wave's next chunk, 384 partial slabs
    (2, 64, 96, 8, 32, 3, 2),         # 48 output columns: not in 64-pixel chunks -> the pixel-axis GEMM
    # model.1 of the s scale (3x3 / s2, 32 -> 64): block-cooperative chunk stream, the 18 accumulator tiles dealt to four waves (wgrad_s2c32_kernel)
    (2, 128, 128, 32, 64, 3, 2),
    (1, 64, 256, 32, 64, 3, 2),       # two chunks per row; top / bottom / left padding
    (4, 320, 256, 32, 64, 3, 2),      # 1 280 chunks on 512 blocks: the register prefetch of a block's next chunk, 512 partial slabs
    (2, 96, 320, 32, 64, 3, 2),       # 160 output columns = 2.5 chunks: the partial last chunk of a row (the s scale at 640 x 640)
    (3, 8, 32, 16, 24, 3, 1),         # n-scale widths
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_wgrad(case, cuda_device):
    from defectdetection_viaobjectdetection_amd import _capi
    B, H, W, cin, cout, k, s = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=k // 2)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).contiguous()            # KRSC
    d_x, d_dy = nhwc16(x, cuda_device), nhwc16(dy, cuda_device)
    d_dw = torch.full((cout, k, k, cin), float("nan"), dtype=torch.float32, device=cuda_device)
    _capi.check(_capi.lib.m355_conv2d_wgrad(_p(d_x), _p(d_dy), B, H, W, cin, cout, k, s, _p(d_dw),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = d_dw.cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) <= 1e-3
    d_dw2 = torch.full_like(d_dw, float("nan"))                 # split-K slabs added in split order: the same bits again
    _capi.check(_capi.lib.m355_conv2d_wgrad(_p(d_x), _p(d_dy), B, H, W, cin, cout, k, s, _p(d_dw2),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(d_dw, d_dw2)


@pytest.mark.parametrize("B,H,W,Cc,acc", [(3, 20, 20, 32, 0), (2, 13, 17, 16, 1), (2, 5, 4, 8, 0)])
def test_sppf_pool_backward_matches_autograd(B, H, W, Cc, acc, cuda_device):
    """The gather kernel against torch autograd through three F.max_pool2d(5, 1, 2) on the same fp16 values.
    Quantised inputs force many exact ties inside the windows: the argmax rule (first maximum in row-major order) matters."""
    from defectdetection_viaobjectdetection_amd import _capi as capi
    g = torch.Generator().manual_seed(5)
    a16 = (torch.randn((B, H, W, Cc), generator=g) * 2).round().div(2).half()            # values on a 0.5 grid: ties everywhere
    a = a16.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y1 = F.max_pool2d(a, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    gy16 = torch.randn((B, H, W, 3 * Cc), generator=g).half()
    gyn = gy16.float().permute(0, 3, 1, 2)
    (ga_ref,) = torch.autograd.grad((y1, y2, y3), a, (gyn[:, :Cc], gyn[:, Cc:2 * Cc], gyn[:, 2 * Cc:]))
    ga_ref = ga_ref.permute(0, 2, 3, 1)
    ycat = torch.cat((y1, y2, y3), 1).permute(0, 2, 3, 1).half().contiguous()
    old = torch.randn((B, H, W, Cc), generator=g).half()
    d_a, d_y, d_gy, d_ga = a16.to(cuda_device), ycat.to(cuda_device), gy16.to(cuda_device), old.clone().to(cuda_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib.m355_sppf_pool_bwd_launch(_p(d_a), H * W * Cc, Cc, _p(d_y), H * W * 3 * Cc, 3 * Cc, _p(d_gy), H * W * 3 * Cc, 3 * Cc,
                                                  _p(d_ga), H * W * Cc, Cc, B, H, W, Cc, acc, st))
    torch.cuda.synchronize()
    want = (old.float() + ga_ref.half().float()).half() if acc else ga_ref.half()
    assert torch.equal(d_ga.cpu(), want), float((d_ga.cpu().float() - want.float()).abs().max())


@pytest.mark.parametrize("f16,nb,rows,cols,ld", [(0, 4, 1600, 97, 97), (0, 64, 100, 97, 97), (0, 3, 37, 145, 160), (0, 1, 5, 256, 256),
                                                  (1, 1, 51200, 64, 64), (1, 2, 333, 32, 96), (1, 5, 7, 512, 512)])
def test_column_sums_match_torch_and_repeat_bit_for_bit(f16, nb, rows, cols, ld, cuda_device):
    """Bias gradients: m355_colsum_launch against `x.float().sum(rows)` on a (nb, rows, cols) view of a wider,
    batch-strided buffer -- fp32 tolerance (another summation order), the second run the same bits (fixed order)."""
    from defectdetection_viaobjectdetection_amd import _capi as capi
    g = torch.Generator().manual_seed(11)
    A = rows + 9                                                           # the view is rows [4, 4 + rows) of each batch entry
    full = torch.randn((nb, A, ld), generator=g)
    full = (full.half() if f16 else full).to(cuda_device)
    view = full[:, 4:4 + rows, :cols]
    ws = torch.empty(int(capi.lib.m355_colsum_workspace_floats(nb, cols)), device=cuda_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for _ in range(2):
        out = torch.full((cols,), float("nan"), device=cuda_device)
        capi.check(capi.lib.m355_colsum_launch(full.data_ptr() + 4 * ld * full.element_size(), f16, nb, A * ld, rows, ld, cols, _p(ws),
                                               _p(out), st))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    want = view.double().sum((0, 1)).cpu()
    scale = view.double().abs().sum((0, 1)).cpu()
    assert torch.equal(outs[0], outs[1])
    assert float(((outs[0].double() - want).abs() / scale).max()) < 2e-6
    # invalid shapes are refused, not launched
    assert capi.lib.m355_colsum_launch(_p(full), f16, nb, A * ld, rows, ld, 4096 if f16 else 300, _p(ws), _p(out), st) != 0


@pytest.mark.parametrize("B,H,W,Cc,acc", [(2, 20, 20, 64, 0), (3, 7, 5, 24, 1), (1, 1, 1, 8, 1)])
def test_upsample_backward_equals_the_torch_expression(B, H, W, Cc, acc, cuda_device):
    """Nearest-2x upsample backward on channel slices of wider buffers (the concat tensors of the neck) against
    `g.reshape(B, H, 2, W, 2, C).float().sum((2, 4))` stored / added as fp16.  Values on a 2^-6 grid: the fp32 sum of four is exact
    whatever its order, so the comparison is bit for bit; a second case with normal values bounds the order effect to one fp16 ulp."""
    from defectdetection_viaobjectdetection_amd import _capi as capi
    gen = torch.Generator().manual_seed(3)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for exact in (True, False):
        gfull = torch.randn((B, 2 * H, 2 * W, Cc + 16), generator=gen)
        gfull = ((gfull * 64).round() / 64 if exact else gfull).half()
        dfull = torch.randn((B, H, W, Cc + 8), generator=gen).half()
        gs = gfull[..., 8:8 + Cc].reshape(B, H, 2, W, 2, Cc).float().sum((2, 4))
        want = dfull.clone()
        want[..., 8:] = (dfull[..., 8:].float() + gs.half().float()).half() if acc else gs.half()
        d_g, d_d = gfull.to(cuda_device), dfull.clone().to(cuda_device)
        capi.check(capi.lib.m355_upsample2x_bwd_launch(d_g.data_ptr() + 16, 4 * H * W * (Cc + 16), Cc + 16, d_d.data_ptr() + 16,
                                                       H * W * (Cc + 8), Cc + 8, B, H, W, Cc, acc, st))
        torch.cuda.synchronize()
        got = d_d.cpu()
        assert torch.equal(got[..., :8], dfull[..., :8])                     # channels outside the slice untouched
        if exact:
            assert torch.equal(got, want)
        else:
            assert float((got.float() - want.float()).abs().max()) <= 2 ** -9 * float(want.float().abs().max())


@pytest.mark.parametrize("npx", [1, 3, 4, 1027, 64 * 64 * 2])
def test_input_conversion_equals_float_div_255_half(npx, cuda_device):
    """uint8 pixels -> the 8-channel fp16 input rows of the training stem: every byte value, bit for bit against
    `(u8.float() / 255).half()` computed on the host (IEEE division), channels 3..7 zero; also from a source pointer that is not
    4-byte aligned."""
    from defectdetection_viaobjectdetection_amd import _capi as capi
    gen = torch.Generator().manual_seed(9)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shift in (0, 1):
        src = torch.randint(0, 256, (npx * 3 + shift,), generator=gen, dtype=torch.uint8)
        if npx * 3 >= 256:
            src[shift:shift + 256] = torch.arange(256, dtype=torch.uint8)
        d_src = src.to(cuda_device)
        out = torch.full((npx, 8), float("nan"), dtype=torch.float16, device=cuda_device)
        capi.check(capi.lib.m355_u8_to_f16x8_launch(d_src.data_ptr() + shift, _p(out), npx, st))
        torch.cuda.synchronize()
        want = torch.zeros((npx, 8), dtype=torch.float16)
        want[:, :3] = torch.from_numpy((src[shift:].view(npx, 3).numpy().astype(np.float32) / np.float32(255.0)).astype(np.float16))
        assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("B,H,W,c,acc", [(2, 16, 16, 16, False), (1, 20, 28, 64, True), (3, 10, 10, 8, False), (2, 40, 40, 128, True)])
def test_adown_pooling_front_forward_and_backward_match_torch(B, H, W, c, acc, cuda_device):
    """ADown's pooling front (upstream ADown.forward: avg_pool2d(x, 2, 1, 0), chunk, max_pool2d(., 3, 2, 1)) as the two HIP launches of
    the v9c training graph, against torch on fp32 copies of the same fp16 input -- values, the stored argmax (through the backward:
    tie-heavy inputs, quantised to a few levels) and the gathered input gradient; store and accumulate forms."""
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(B * 1000 + H * W + c)
    x = (torch.randint(-3, 4, (B, 2 * c, H, W), generator=g).float() / 2).half()      # few distinct values: many equal maxima per window
    xf = x.float().requires_grad_(True)
    t = F.avg_pool2d(xf, 2, 1, 0)
    p1, p2 = t[:, :c], F.max_pool2d(t[:, c:], 3, 2, 1)
    g1 = torch.randn(p1.shape, generator=g).half().float()
    g2 = torch.randn(p2.shape, generator=g).half().float()
    (gx,) = torch.autograd.grad((p1, p2), xf, (g1, g2))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_x = nhwc16(x.float(), cuda_device)
    Ho, Wo = H // 2, W // 2
    d_p1 = torch.empty((B, H - 1, W - 1, c), dtype=torch.float16, device=cuda_device)
    d_p2 = torch.empty((B, Ho, Wo, c), dtype=torch.float16, device=cuda_device)
    d_arg = torch.empty((B, Ho, Wo, c), dtype=torch.uint8, device=cuda_device)
    _capi.check(_capi.lib.m355_adown_fwd_launch(_p(d_x), H * W * 2 * c, 2 * c, _p(d_p1), (H - 1) * (W - 1) * c, c, _p(d_p2), Ho * Wo * c, c,
                                                _p(d_arg), B, H, W, c, st))
    assert torch.equal(d_p1.cpu(), p1.detach().permute(0, 2, 3, 1).half())
    assert torch.equal(d_p2.cpu(), p2.detach().permute(0, 2, 3, 1).half())
    old = torch.randn(B, H, W, 2 * c, generator=g).half()
    d_gx = old.to(cuda_device).clone() if acc else torch.full((B, H, W, 2 * c), float("nan"), dtype=torch.float16, device=cuda_device)
    d_g1, d_g2 = nhwc16(g1, cuda_device), nhwc16(g2, cuda_device)        # (named: a temporary inside the argument list is freed at once)
    _capi.check(_capi.lib.m355_adown_bwd_launch(_p(d_g1), (H - 1) * (W - 1) * c, c, _p(d_g2), Ho * Wo * c, c,
                                                _p(d_arg), _p(d_gx), H * W * 2 * c, 2 * c, B, H, W, c, 1 if acc else 0, st))
    want = gx.permute(0, 2, 3, 1).half()
    if acc:
        want = (old.float() + want.float()).half()
    got = d_gx.cpu()
    assert torch.isfinite(got).all()
    # sums of <= 16 fp32 terms in a different order, then one fp16 rounding: equal up to an ulp on a handful of elements
    assert rel_l2(got.float(), want.float()) <= 2e-4
    assert float((got.float() - want.float()).abs().max()) <= 4e-3 * float(want.float().abs().max())


@pytest.mark.parametrize("npix,Cc,ldy", [(1000, 64, 64), (777, 128, 320), (64 * 400, 256, 256)])
def test_repconv_tail_silu_of_the_branch_sum_matches_torch(npix, Cc, ldy, cuda_device):
    """RepConvN's tail y = SiLU(a + b) and its backward g = dy * SiLU'(a + b) (one gradient for both branches), against the torch
    expressions the v9c training graph used before (fp16 sum kept, fp32 SiLU, one rounding)."""
    from defectdetection_viaobjectdetection_amd import _capi
    g = torch.Generator().manual_seed(npix + Cc)
    a = (torch.randn(npix, Cc, generator=g) * 2).half().to(cuda_device)
    b = (torch.randn(npix, Cc, generator=g) * 2).half().to(cuda_device)
    dy_full = torch.randn(npix, ldy, generator=g).half().to(cuda_device)
    y_full = torch.zeros(npix, ldy, dtype=torch.float16, device=cuda_device)
    v = torch.empty(npix, Cc, dtype=torch.float16, device=cuda_device)
    gg = torch.empty(npix, Cc, dtype=torch.float16, device=cuda_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    off = ldy - Cc
    _capi.check(_capi.lib.m355_addsilu_fwd_launch(_p(a), _p(b), _p(v), C.c_void_p(y_full.data_ptr() + 2 * off), npix, ldy, Cc, st))
    _capi.check(_capi.lib.m355_addsilu_bwd_launch(_p(v), C.c_void_p(dy_full.data_ptr() + 2 * off), ldy, _p(gg), npix, Cc, st))
    torch.cuda.synchronize()
    v_ref = torch.add(a, b)
    assert torch.equal(v, v_ref)
    vf = v_ref.float()
    y_ref = F.silu(vf)
    sig = torch.sigmoid(vf)
    g_ref = dy_full[:, off:].float() * (sig * (1.0 + vf * (1.0 - sig)))
    assert float((y_full[:, :off]).abs().max()) == 0.0 if off else True
    # v_exp / v_rcp sigmoid (1 ulp in fp32) then one fp16 rounding: at most one fp16 ulp apart
    assert rel_l2(y_full[:, off:].float().cpu(), y_ref.cpu()) <= 3e-4
    assert rel_l2(gg.float().cpu(), g_ref.cpu()) <= 3e-4
