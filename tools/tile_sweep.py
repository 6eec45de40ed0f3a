"""Time representative layers on the im2col / halo tile configurations.
Run under rocprofv3 --kernel-trace (output CSV), then `python tools/tile_sweep.py --parse <kernel_trace.csv>`:
every (shape, tile) launches its kernel five times (dbg bit 128) and the parser takes the fastest of each group."""
import ctypes as C, sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # B, H, W, cin, cout, k, stride
    (32, 20, 20, 256, 256, 3, 1), (32, 20, 20, 512, 64, 3, 1), (32, 40, 40, 256, 256, 3, 2),
    (32, 160, 160, 64, 128, 3, 2), (32, 320, 320, 32, 64, 3, 2), (32, 80, 80, 128, 256, 3, 2), (32, 40, 40, 256, 512, 3, 2),
    (32, 80, 80, 384, 128, 1, 1), (32, 40, 40, 768, 256, 1, 1), (32, 80, 80, 256, 128, 1, 1), (32, 20, 20, 768, 512, 1, 1)]
TILES = (0, 1, 3)
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "conv_igemm" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for sh in SHAPES:
        out = []
        for t in TILES:
            grp = rows[i:i + 5]; i += 5
            out.append(f"tile {t}: {min(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in grp) / 1e3:7.1f} us")
        print(sh, "  ".join(out))
    sys.exit(0)
import torch
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
def run(B, H, W, cin, cout, k, stride, tile):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, k, k) * 0.05; b = torch.zeros(cout)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(B, Ho, Ho, cout, device='cuda', dtype=torch.float16)
    _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, k, stride, 1, P(None), P(y), 0, tile | (128 << 8),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
for sh in SHAPES:
    for tile in TILES:
        run(*sh, tile)
