"""In-kernel s_memtime / s_memrealtime stamps of the halo conv: prologue / main loop / epilogue cycles per block,
in-kernel clock, MFMA-cycle share of the main loop.  Usage: python tools/stamps_halo.py [tile] [dbg]"""
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M355_STAMPS"] = "/tmp/stamps.bin"
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 18
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def run(B, H, W, cin, cout, k, mfma_per_wave):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * 0.05; b = torch.zeros(cout)
    y = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    for _ in range(2):
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, k, 1, 1, P(None), P(y), 0, tile | (dbg << 8),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    raw = np.fromfile("/tmp/stamps.bin", dtype=np.uint64)
    s2 = raw[1 << 19:(1 << 19) + (1 << 18)].reshape(-1, 4).astype(np.int64)
    s3 = raw[(1 << 19) + (1 << 18):].astype(np.int64)[:len(s2)]
    okm = (s2[:, 0] > 0) & (s3 > 0)
    if okm.any():
        print(f"   main end of tile 1 -> loop top of tile 2 (decode + setup + prologue issue) med {np.median((s3 - s2[:, 3])[okm]):.0f} cycles")
    s2 = s2[s2[:, 0] > 0]
    if len(s2):
        print(f"   second tile med: epilogue(prev) {np.median(s2[:, 1] - s2[:, 0]):.0f}  wait+barrier+first reads {np.median(s2[:, 2] - s2[:, 1]):.0f}"
              f"  main loop {np.median(s2[:, 3] - s2[:, 2]):.0f} cycles")
    s = raw[:1 << 19].reshape(-1, 8)
    s = s[s[:, 0] > 0].astype(np.int64)
    pro = s[:, 1] - s[:, 0]; main = s[:, 2] - s[:, 1]; epi = s[:, 3] - s[:, 2]; tot = s[:, 3] - s[:, 0]
    rt = (s[:, 5] - s[:, 4]).clip(1)
    clk = np.median(tot / rt) * 100e6
    span_rt = (s[:, 5].max() - s[:, 4].min()) / 100e6
    print(f"{(B, H, W, cin, cout)} tile {tile} dbg {dbg}: blocks {len(s)} kernel span {span_rt * 1e6:.1f} us, in-kernel clock {clk / 1e9:.2f} GHz")
    if s[:, 6].max() > 0:
        nt = s[:, 6].clip(1)
        print(f"   persistent: tiles/block med {np.median(nt):.0f}, cycles per tile (block lifetime / tiles) med {np.median(tot / nt):.0f};"
              f" MFMA share 2 x {mfma_per_wave * 16} / that = {2 * mfma_per_wave * 16 / np.median(tot / nt):.2f}")
    print(f"   cycles med: prologue {np.median(pro):.0f} main {np.median(main):.0f} epilogue {np.median(epi):.0f} total {np.median(tot):.0f};"
          f" MFMA cycles/wave {mfma_per_wave * 16} -> x2 waves/SIMD = {2 * mfma_per_wave * 16 / np.median(main):.2f} of the main loop")
if tile == 25:   # slab kernel: 20x20 maps, 72 steps x 16 MFMAs per wave
    run(32, 20, 20, 256, 256, 3, 72 * 16)
    sys.exit(0)
run(32, 160, 160, 128, 128, 3, 18 * 32)
if tile != 19:
    run(32, 40, 40, 128, 128, 3, 18 * 32)
else:
    run(32, 80, 80, 128, 128, 3, 18 * 32)
