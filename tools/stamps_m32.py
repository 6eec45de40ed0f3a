"""In-kernel s_memtime stamps of the 32x32x16 halo conv (conv3x3_m32.hip): prologue / main loop / epilogue cycles per block,
in-kernel clock, MFMA-cycle share.  Usage: python tools/stamps_m32.py"""
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M355_STAMPS"] = "/tmp/stamps.bin"
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B, H, W, cin, cout, tile, dbg=0):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5); b = torch.zeros(cout)
    y = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    for _ in range(3):
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, 3, 1, 1, P(None), P(y), 0, tile | (dbg << 8),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    raw = np.fromfile("/tmp/stamps.bin", dtype=np.uint64)
    s = raw[:1 << 19].reshape(-1, 8)
    s = s[s[:, 0] > 0].astype(np.int64)
    pro = s[:, 1] - s[:, 0]; tot = s[:, 3] - s[:, 0]; nt = s[:, 6].clip(1)
    rt = (s[:, 5] - s[:, 4]).clip(1)
    clk = np.median(tot / rt) * 100e6
    span = (s[:, 5].max() - s[:, 4].min()) / 100e6
    start_spread = (s[:, 4].max() - s[:, 4].min()) / 100e6
    mf = (cin // 64) * 9 * 16 * 32     # MFMA pipe cycles per wave
    print(f"{(B, H, W, cin, cout)} tile {tile} dbg {dbg}: blocks {len(s)}, kernel span {span * 1e6:.1f} us (block starts spread over {start_spread * 1e6:.1f} us), "
          f"in-kernel clock {clk / 1e9:.2f} GHz")
    per_tile = (tot - pro) / nt
    print(f"   tiles per block med {np.median(nt):.0f} (max {nt.max()}); cycles med: prologue {np.median(pro):.0f}, block life {np.median(tot):.0f}, per tile "
          f"(main + epilogue) {np.median(per_tile):.0f}; MFMA pipe cycles per wave and tile {mf}: x2 waves/SIMD = {2 * mf / np.median(per_tile):.2f} of a "
          f"tile's time, {2 * mf * np.median(nt) / np.median(tot):.2f} of the block's life")
for dbg in (0, 1, 2, 3, 4, 7):
    run(32, 80, 80, 128, 128, 27, dbg)
run(32, 40, 40, 128, 128, 27)
run(32, 80, 80, 64, 64, 29)
run(32, 40, 40, 256, 224, 27)
