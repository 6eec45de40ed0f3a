import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def conv(x, w, b, k, stride, act, res, tile):
    B, H, W, cin = x.shape
    cout = w.shape[0]
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(B, Ho, Ho, cout, device='cuda', dtype=torch.float16)
    _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, k, stride, act, P(res), P(y), 0, tile,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return y
torch.manual_seed(0)
for (H, cin, cout, k, stride, tile, use_res) in [(80, 256, 128, 1, 1, -1, False), (80, 64, 64, 3, 1, -1, True), (40, 128, 128, 3, 1, -1, True),
                                                 (20, 256, 256, 3, 1, -1, False), (160, 32, 64, 3, 2, -1, False), (20, 512, 256, 1, 1, -1, False)]:
    x = torch.randn(32, H, H, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5); b = torch.randn(cout) * 0.1
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(32, Ho, Ho, cout, device='cuda').half() if use_res else None
    y = conv(x, w, b, k, stride, 1, res, tile)
    y3 = conv(x[5:8].contiguous(), w, b, k, stride, 1, None if res is None else res[5:8].contiguous(), tile)
    d = (y[5:8].float() - y3.float()).abs()
    print((H, cin, cout, k, stride), "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel())
print("run-to-run determinism, same batch:")
for (H, cin, cout, k, stride) in [(80, 256, 128, 1, 1), (20, 512, 256, 1, 1), (20, 256, 256, 3, 1)]:
    x = torch.randn(32, H, H, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5); b = torch.randn(cout) * 0.1
    ys = [conv(x, w, b, k, stride, 1, None, -1) for _ in range(4)]
    print((H, cin, cout, k), [int((ys[0] != y).sum()) for y in ys[1:]])
    y3 = [conv(x[5:8].contiguous(), w, b, k, stride, 1, None, -1) for _ in range(3)]
    print("   sub-batch vs full:", [int((ys[0][5:8] != y).sum()) for y in y3], " sub-batch run-to-run:", int((y3[0] != y3[1]).sum()))
print("forced tiles, sub-batch vs full (fast epilogue):")
H, cin, cout, k, stride = 80, 256, 128, 1, 1
x = torch.randn(32, H, H, cin, device='cuda').half()
w = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5); b = torch.randn(cout) * 0.1
ref = {t: conv(x, w, b, k, stride, 1, None, t) for t in (0, 1, 3)}
for t in (0, 1, 3):
    y3 = conv(x[5:8].contiguous(), w, b, k, stride, 1, None, t)
    print(" tile", t, "sub vs full same tile:", int((ref[t][5:8] != y3).sum()), " full tile", t, "vs full tile 0:", int((ref[t] != ref[0]).sum()))
idx = (ref[1] != ref[0]).nonzero()[:3]
for i in idx:
    i = tuple(int(v) for v in i)
    print("   elem", i, float(ref[0][i]), float(ref[1][i]))
# the same with act = 0
r0 = conv(x, w, b, k, stride, 0, None, 0); r1 = conv(x, w, b, k, stride, 0, None, 1)
print(" act=0: tile 1 vs tile 0:", int((r0 != r1).sum()))
