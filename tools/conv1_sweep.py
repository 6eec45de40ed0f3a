"""1x1 layer shapes of YOLOv8s-seg at batch 32 on the implicit-GEMM kernel, one-tile-per-block vs persistent (dbg bit 64).
Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/conv1_sweep.py`, then
`python tools/conv1_sweep.py --parse DIR/**/t_kernel_trace.csv`: every (shape, variant) launches five times (dbg bit 128);
the parser reports the fastest with GB/s of algorithmic bytes and the fraction of 8 TB/s."""
import ctypes as C, sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # B, H, W, cin, cout (layer)
    (32, 160, 160, 96, 64),    # model.2.cv2
    (32, 80, 80, 256, 128),    # model.4.cv2
    (32, 80, 80, 384, 128),    # model.15.cv1 (without the read-through)
    (32, 80, 80, 192, 128),    # model.15.cv2
    (32, 40, 40, 256, 256),    # model.6.cv1
    (32, 40, 40, 512, 256),    # model.6.cv2
    (32, 40, 40, 768, 256),    # model.12.cv1
    (32, 40, 40, 384, 256),    # model.12.cv2 / 18.cv1 / 18.cv2
    (32, 20, 20, 512, 512),    # model.8.cv1
    (32, 20, 20, 768, 512),    # model.8.cv2 / 21.*
    (32, 20, 20, 1024, 512),   # model.9.cv2
    (32, 20, 20, 512, 256),    # model.9.cv1
]
VARIANTS = {0: "tile/block", 64: "persistent"}
EXTRA = [int(v) for v in os.environ.get("CONV1_VARIANTS", "").split(",") if v]
for v in EXTRA:
    VARIANTS[v] = f"dbg{v}"
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "conv_igemm" in r["Kernel_Name"] or "conv1x1" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for sh in SHAPES:
        B, H, W, cin, cout = sh
        by = B * H * W * (cin + cout) * 2 + cin * cout * 2
        out = []
        for v, name in VARIANTS.items():
            grp = rows[i:i + 5]; i += 5
            us = min(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in grp) / 1e3
            out.append(f"{name}: {us:6.1f} us {by / us / 1e3:6.0f} GB/s ({by / us / 1e3 / 8000:.2f})")
        print(sh, " | ".join(out))
    sys.exit(0)
import torch
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
for sh in SHAPES:
    B, H, W, cin, cout = sh
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, 1, 1) * (1.0 / cin ** 0.5); b = torch.randn(cout) * 0.1
    y = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    M = B * H * W
    tile = 1 if (cout <= 64 or M * ((cout + 127) // 128) // 128 < 300) else 0    # conv_pick_tile
    for v in VARIANTS:
        _capi.check(_capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, 1, 1, 1, P(None), P(y), 0, tile | ((128 | v) << 8),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
