"""Per-op parity of the HIP kernels (through the C-ABI) against plain PyTorch fp32 CPU references.

Tolerance: fp16 in / fp32 accumulate conv output rel-L2 <= 1e-3 vs the fp32 reference
evaluated on the same fp16-rounded inputs and weights; pooling / upsample are bit-exact.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from defectdetection_viaobjectdetection_amd import _capi
    return _capi


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def nhwc16(x_nchw, dev):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(dev)


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


CONV_CASES = [
    # B, H, W, cin, cout, k, s, act, res, tile
    (2, 20, 20, 64, 64, 3, 1, 1, False, -1),
    (2, 20, 20, 64, 64, 3, 1, 1, True, -1),
    (1, 40, 40, 32, 32, 3, 1, 1, True, -1),
    (2, 24, 16, 128, 128, 3, 1, 1, False, -1),
    (2, 24, 16, 128, 256, 3, 2, 1, False, -1),
    (1, 17, 23, 64, 128, 3, 2, 1, False, -1),      # odd sizes: ragged pixel tiles + stride-2 padding
    (3, 20, 20, 768, 256, 1, 1, 1, False, -1),
    (2, 16, 16, 96, 64, 1, 1, 1, False, -1),        # K not a multiple of 64
    (2, 16, 16, 48, 48, 3, 1, 1, True, -1),         # m-scale widths
    (2, 16, 16, 16, 16, 3, 1, 1, False, -1),        # n-scale widths
    (2, 20, 20, 128, 128, 3, 1, 1, False, 1),       # forced 64x128 tile
    (2, 20, 20, 128, 128, 3, 1, 1, False, 2),       # forced 32x256 tile
    (2, 20, 20, 128, 128, 3, 1, 1, False, 3),       # forced 64x256 tile
    (2, 20, 20, 64, 64, 1, 1, 0, False, 0),         # forced 128x128 tile with Cout=64
    (4, 80, 80, 128, 128, 3, 1, 1, True, -1),       # many tiles
    # 8-wave im2col tile (5: 64 ch x 128 px, 8 waves)
    (2, 24, 16, 128, 128, 3, 1, 1, True, 5),
    (1, 17, 23, 64, 128, 3, 2, 1, False, 5),
    (2, 20, 20, 256, 64, 3, 1, 1, True, 5),
    (2, 16, 16, 48, 48, 1, 1, 0, False, 5),
    # halo-tile 3x3 kernel (tile id 16)
    (2, 32, 32, 64, 64, 3, 1, 1, False, 16),        # single chunk, 64-ch variant
    (2, 32, 48, 64, 64, 3, 1, 1, True, 16),
    (2, 32, 32, 128, 128, 3, 1, 1, True, 16),       # two chunks, 128-ch variant
    (1, 80, 80, 128, 224, 3, 1, 1, False, 16),      # ragged channel tile (224 = 128 + 96)
    (2, 48, 32, 192, 64, 3, 1, 0, False, 16),       # 3 chunks, no act
    (2, 72, 56, 64, 64, 3, 1, 1, True, 16),         # partial spatial tiles (72 = 4.5 x 16, 56 = 3.5 x 16)
    (1, 160, 160, 128, 128, 3, 1, 1, False, 16),
    # both halo variants explicitly: 17 = 8 waves / 16x16 px, 18 = 4 waves / 8x16 px
    (2, 32, 32, 128, 128, 3, 1, 1, True, 17),
    (2, 72, 56, 64, 64, 3, 1, 1, True, 17),
    (1, 80, 80, 128, 224, 3, 1, 1, False, 17),
    (2, 32, 32, 128, 128, 3, 1, 1, True, 18),
    (2, 72, 56, 64, 64, 3, 1, 1, True, 18),
    (1, 80, 80, 128, 224, 3, 1, 1, False, 18),
    # wide halo kernel (19): 128 ch x 16x16 px, 32-deep K steps
    (2, 32, 32, 128, 128, 3, 1, 1, True, 19),       # 4 chunks, residual
    (1, 80, 80, 128, 224, 3, 1, 1, False, 19),      # ragged channel tile
    (2, 72, 56, 64, 128, 3, 1, 1, True, 19),        # 2 chunks, partial spatial tiles
    (1, 160, 160, 128, 128, 3, 1, 1, False, 19),
    (3, 48, 32, 192, 256, 3, 1, 0, False, 19),      # 6 chunks, two channel tiles, no act
    # narrow weights-stationary kernel (20): Cin = Cout = 32
    (2, 32, 32, 32, 32, 3, 1, 1, True, 20),         # residual, full tiles
    (3, 72, 56, 32, 32, 3, 1, 1, False, 20),        # partial spatial tiles
    (1, 160, 160, 32, 32, 3, 1, 0, False, 20),      # several tiles per block is exercised by the engine tests; no act
    # 1x1 with the weights in registers (32, conv1x1_wreg.hip): K <= 512, Cout % 128 == 0, flat pixel axis
    (2, 80, 80, 256, 128, 1, 1, 1, False, 32),      # C2f.cv2 at 80 x 80: 128-pixel tiles, four channel blocks x two pixel halves
    (1, 80, 80, 192, 128, 1, 1, 1, False, 32),      # K = 192: 3-bit chunk swizzle, three tile buffers
    (2, 40, 40, 256, 256, 1, 1, 1, False, 32),      # eight channel blocks, 64-pixel tiles
    (3, 40, 40, 384, 256, 1, 1, 0, False, 32),      # K = 384, no activation, two tile buffers
    (4, 20, 20, 512, 512, 1, 1, 1, False, 32),      # two channel tiles of 256: blocks cross the channel-tile boundary
    (24, 40, 40, 512, 256, 1, 1, 1, False, 32),     # 600 tiles over 256 blocks: 2 or 3 tiles per block, K = 512
    (2, 8, 8, 128, 128, 1, 1, 1, False, 32),        # one 128-pixel tile, K = 128
    (3, 40, 40, 256, 128, 1, 1, 1, False, 32),      # 4800 pixels = 37.5 tiles of 128: the partial last pixel tile is masked
    (3, 20, 20, 512, 512, 1, 1, 1, False, 32),      # 1200 pixels = 18.75 tiles of 64, two channel tiles: a partial tile mid-walk
    (1, 10, 10, 384, 256, 1, 1, 1, False, 32),      # 100 pixels: two tiles, two buffers
    # slab kernel for narrow maps (25): R full-width rows x 64 channels, linear pixel groups
    (2, 20, 20, 256, 256, 3, 1, 1, True, 25),       # the 20x20 C2f layers: two slabs of 10 rows, 12.5 groups
    (3, 20, 20, 512, 224, 3, 1, 0, False, 25),      # fused head-level convs: ragged channel tile (224 = 3 x 64 + 32)
    (2, 10, 10, 64, 64, 3, 1, 1, False, 25),        # one slab of 100 pixels = 6.25 groups (waves 2 and 3 mostly idle)
    (1, 26, 26, 64, 128, 3, 1, 1, True, 25),        # widest map: R = 9, last slab 8 rows
    (2, 13, 7, 128, 64, 3, 1, 1, False, 25),        # odd sizes, groups straddling up to three rows
    (300, 20, 20, 64, 64, 3, 1, 1, False, 25),      # 600 tiles on 512 slots: blocks walk two tiles
    # v_mfma_f32_32x32x16_f16 halo kernel (conv3x3_m32.hip): 27 = 128 ch x 8 rows, 28 = 64 ch x 16 rows, 29 = 64 ch x 8 rows,
    # 26 = by shape
    (2, 32, 32, 128, 128, 3, 1, 1, True, 27),       # two chunks (both patch buffers), residual
    (1, 80, 80, 128, 224, 3, 1, 1, False, 27),      # ragged channel tile (224 = 128 + 96)
    (2, 40, 40, 128, 128, 3, 1, 1, True, 27),       # 2.5 column tiles
    (2, 72, 56, 64, 128, 3, 1, 1, True, 27),        # one chunk (single patch buffer), partial tiles in both directions
    (3, 48, 32, 192, 256, 3, 1, 0, False, 27),      # three chunks (odd: the unpaired tail chunk), two channel tiles, no act
    (2, 40, 40, 256, 224, 3, 1, 1, False, 27),      # four chunks
    (1, 8, 16, 64, 128, 3, 1, 1, False, 27),        # a single tile: every image border inside one patch
    (2, 80, 80, 64, 64, 3, 1, 1, True, 28),
    (2, 72, 56, 64, 64, 3, 1, 1, True, 28),         # partial spatial tiles
    (1, 80, 80, 128, 96, 3, 1, 0, False, 28),       # two chunks, two channel tiles, the second half full (96 = 64 + 32)
    (2, 40, 40, 64, 64, 3, 1, 1, True, 29),
    (2, 24, 40, 192, 64, 3, 1, 1, False, 29),       # three chunks
    (1, 15, 30, 64, 72, 3, 1, 1, True, 29),         # odd sizes, Cout not a multiple of 16
    (6, 160, 160, 64, 64, 3, 1, 1, False, 26),      # by shape
    (2, 40, 40, 256, 256, 3, 1, 1, True, 26),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd(case, cuda_device):
    capi = _lib()
    B, H, W, cin, cout, k, s, act, use_res, tile = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xh = x.half().float()
    wh = w.half().float()
    y_ref = F.conv2d(xh, wh, b, stride=s, padding=k // 2)
    if act:
        y_ref = F.silu(y_ref)
    res = None
    if use_res:
        res = torch.randn(y_ref.shape, generator=g).half().float()
        y_ref = y_ref + res
    dx = nhwc16(x, cuda_device)
    dres = nhwc16(res, cuda_device) if use_res else None
    Ho, Wo = y_ref.shape[2], y_ref.shape[3]
    dy = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.float16, device=cuda_device)
    rc = capi.lib.m355_conv2d_fwd(_p(dx), B, H, W, cin, _p(w.contiguous()), _p(b.contiguous()), cout, k, s, act,
                                  _p(dres), _p(dy), 0, tile, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    capi.check(rc)
    y = dy.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(y).all()
    err = rel_l2(y, y_ref)
    assert err <= 1e-3, f"rel-L2 {err}"


def test_conv2d_f32_out_small_cout(cuda_device):
    """Head-final 1x1 convs: fp32 output, Cout = nc = 1 and 64, no activation."""
    capi = _lib()
    for cout in (1, 3, 64):
        B, H, W, cin = 2, 20, 20, 128
        g = torch.Generator().manual_seed(cout)
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        b = torch.randn(cout, generator=g)
        y_ref = F.conv2d(x.half().float(), w.half().float(), b)
        dx = nhwc16(x, cuda_device)
        dy = torch.full((B, H, W, cout), float("nan"), dtype=torch.float32, device=cuda_device)
        rc = capi.lib.m355_conv2d_fwd(_p(dx), B, H, W, cin, _p(w.contiguous()), _p(b.contiguous()), cout, 1, 1, 0,
                                      _p(None), _p(dy), 1, -1, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        capi.check(rc)
        y = dy.cpu().permute(0, 3, 1, 2)
        assert rel_l2(y, y_ref) <= 1e-5


@pytest.mark.parametrize("shape", [(2, 20, 12, 128, 128), (2, 16, 32, 128, 128), (3, 8, 16, 64, 256), (1, 12, 16, 128, 48)])
def test_convt2x2(shape, cuda_device):
    """(W % 16 == 0 and Co % 128 == 0 take the fast pixel-shuffle epilogue, the others the generic one)"""
    capi = _lib()
    B, H, W, cin, cout = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y_ref = F.conv_transpose2d(x.half().float(), w.half().float(), b, stride=2)
    dx = nhwc16(x, cuda_device)
    dy = torch.full((B, 2 * H, 2 * W, cout), float("nan"), dtype=torch.float16, device=cuda_device)
    rc = capi.lib.m355_convt2x2_fwd(_p(dx), B, H, W, cin, _p(w.contiguous()), _p(b.contiguous()), cout, _p(dy),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    capi.check(rc)
    y = dy.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(y, y_ref) <= 1e-3


@pytest.mark.parametrize("hw", [(64, 96), (40, 100), (66, 80)])   # W*3 a multiple of 16 (row-staged kernel) or not (gather kernel)
@pytest.mark.parametrize("cout", [16, 32, 48])
def test_stem(cout, hw, cuda_device):
    capi = _lib()
    B, (H, W) = 2, hw
    g = torch.Generator().manual_seed(cout)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    w = torch.randn(cout, 3, 3, 3, generator=g) / 27 ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = img.permute(0, 3, 1, 2).float() / 255.0
    y_ref = F.silu(F.conv2d(x, w, b, stride=2, padding=1))
    dy = torch.full((B, H // 2, W // 2, cout), float("nan"), dtype=torch.float16, device=cuda_device)
    rc = capi.lib.m355_stem_fwd(_p(img.to(cuda_device)), B, H, W, _p(w.contiguous()), _p(b.contiguous()), cout,
                                _p(dy), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    capi.check(rc)
    y = dy.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(y, y_ref) <= 1e-3


def test_sppf_pool_bit_exact(cuda_device):
    capi = _lib()
    B, H, W, Cc = 3, 20, 20, 256
    x = torch.randn(B, Cc, H, W, generator=torch.Generator().manual_seed(1)).half()
    p1 = F.max_pool2d(x.float(), 5, 1, 2)
    p2 = F.max_pool2d(p1, 5, 1, 2)
    p3 = F.max_pool2d(p2, 5, 1, 2)
    ref = torch.cat((p1, p2, p3), 1)
    dx = nhwc16(x.float(), cuda_device)
    dy = torch.empty((B, H, W, 3 * Cc), dtype=torch.float16, device=cuda_device)
    capi.check(capi.lib.m355_sppf_pool(_p(dx), B, H, W, Cc, _p(dy), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    y = dy.float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(y, ref)


def test_upsample_bit_exact(cuda_device):
    capi = _lib()
    B, H, W, Cc = 2, 20, 12, 64
    x = torch.randn(B, Cc, H, W, generator=torch.Generator().manual_seed(2)).half()
    ref = F.interpolate(x.float(), scale_factor=2, mode="nearest")
    dx = nhwc16(x.float(), cuda_device)
    dy = torch.empty((B, 2 * H, 2 * W, Cc), dtype=torch.float16, device=cuda_device)
    capi.check(capi.lib.m355_upsample2x(_p(dx), B, H, W, Cc, _p(dy), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(dy.float().cpu().permute(0, 3, 1, 2), ref)
