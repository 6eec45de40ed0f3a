import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo")
os.environ["M355_STAMPS"] = "/tmp/stamps.bin"
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B, H, W, cin, cout, k, stride, tile, reps=2):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * 0.05; b = torch.zeros(cout)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(B, Ho, Ho, cout, device='cuda', dtype=torch.float16)
    for _ in range(reps):
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, k, stride, 1, P(None), P(y), 0, tile,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    s = np.fromfile("/tmp/stamps.bin", dtype=np.uint64)[:1 << 19].reshape(-1, 8)
    s = s[s[:, 0] > 0].astype(np.int64)
    pro = s[:, 1] - s[:, 0]; main = s[:, 2] - s[:, 1]; epi = s[:, 3] - s[:, 2]; tot = s[:, 3] - s[:, 0]
    span = (s[:, 5].max() - s[:, 4].min()) / 100e6
    nk = (cin * k * k + 63) // 64
    print(f"B={B} {(H, W, cin, cout, k, stride)}: blocks {len(s)} span {span * 1e6:.1f} us; prologue {np.median(pro):.0f} K loop {np.median(main):.0f} ({np.median(main) / nk:.0f}/step) epilogue {np.median(epi):.0f} total {np.median(tot):.0f}")
shapes = [(80, 80, 384, 128), (80, 80, 192, 128), (80, 80, 128, 128), (40, 40, 768, 256), (40, 40, 256, 256), (40, 40, 384, 256),
          (20, 20, 512, 512), (20, 20, 768, 512), (20, 20, 512, 256), (20, 20, 1024, 512), (160, 160, 96, 64), (160, 160, 64, 64)]
for (H, W, ci, co) in shapes:
    run(32, H, W, ci, co, 1, 1, -1 if len(sys.argv) < 2 else int(sys.argv[1]))
