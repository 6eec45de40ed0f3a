"""In-kernel stamps of the implicit-GEMM conv (first tile of each block): prologue / K loop / epilogue cycles."""
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["M355_STAMPS"] = "/tmp/stamps.bin"
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B, H, W, cin, cout, k, stride, tile):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * 0.05; b = torch.zeros(cout)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(B, Ho, Ho, cout, device='cuda', dtype=torch.float16)
    for _ in range(2):
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, k, stride, 1, P(None), P(y), 0, tile,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    s = np.fromfile("/tmp/stamps.bin", dtype=np.uint64)[:1 << 19].reshape(-1, 8)
    s = s[s[:, 0] > 0].astype(np.int64)
    pro = s[:, 1] - s[:, 0]; main = s[:, 2] - s[:, 1]; epi = s[:, 3] - s[:, 2]; tot = s[:, 3] - s[:, 0]
    rt = (s[:, 5] - s[:, 4]).clip(1)
    span = (s[:, 5].max() - s[:, 4].min()) / 100e6
    nk = (cin * k * k + 63) // 64
    print(f"{(B, H, W, cin, cout, k, stride)} tile {tile}: blocks {len(s)} span {span * 1e6:.1f} us clock {np.median(tot / rt) * 0.1:.2f} GHz; "
          f"cycles med: prologue {np.median(pro):.0f} K loop {np.median(main):.0f} ({nk} steps, {np.median(main) / nk:.0f}/step) epilogue {np.median(epi):.0f} total {np.median(tot):.0f}")
run(32, 80, 80, 256, 128, 1, 1, 0)
run(32, 40, 40, 512, 256, 1, 1, 0)
run(32, 80, 80, 128, 128, 1, 1, 0)
run(32, 80, 80, 128, 256, 3, 2, 0)
run(32, 20, 20, 256, 256, 3, 1, 1)
run(32, 160, 160, 32, 32, 3, 1, 2)
