"""3x3 stride-1 layer shapes of YOLOv8s-seg at batch 32 on every 3x3 kernel that accepts them.
Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/conv3_sweep.py`, then
`python tools/conv3_sweep.py --parse DIR/**/t_kernel_trace.csv`: every (shape, tile) launches its kernel five times
(dbg bit 128) and the parser reports the fastest, with TFLOP/s and the fraction of the 2.5 PFLOP/s dense fp16 peak."""
import ctypes as C, sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # B, H, W, cin, cout  (layer)
    (32, 40, 40, 128, 128),   # model.6.m.*, model.12/18.m.*, cv3.1.1
    (32, 80, 80, 64, 64),     # model.4.m.*, model.15.m.*, cv2.0.1
    (32, 80, 80, 128, 128),   # proto.cv1, cv3.0.1
    (32, 80, 80, 128, 224),   # head level 0 first convs (fused)
    (32, 40, 40, 256, 224),   # head level 1 first convs
    (32, 40, 40, 64, 64),     # cv2.1.1
    (32, 160, 160, 64, 64),   # (m-seg-like large map)
]
TILES = {16 + 2: "halo4w", 19: "wide", 27: "m32<128,8>", 28: "m32<64,16>", 29: "m32<64,8>"}
def tiles_for(sh):
    cout = sh[4]
    ts = [18]
    if cout >= 128: ts += [19, 27]
    else: ts += [28, 29]
    return ts
OKFILE = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "conv3_sweep_ok.json")
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import json
    ran = set(tuple(k) for k in json.load(open(OKFILE)))
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "conv3x3" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for sh in SHAPES:
        B, H, W, cin, cout = sh
        fl = 2.0 * B * H * W * cin * cout * 9
        out = []
        for t in tiles_for(sh):
            if (SHAPES.index(sh), t) not in ran:
                continue
            grp = rows[i:i + 5]; i += 5
            us = min(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in grp) / 1e3
            out.append(f"{TILES[t]}: {us:6.1f} us {fl / us / 1e6:6.0f} TF ({fl / us / 1e6 / 2500:.2f})")
        print(sh, " | ".join(out))
    sys.exit(0)
import torch
from defectdetection_viaobjectdetection_amd import _capi
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
ran = []
for sh in SHAPES:
    B, H, W, cin, cout = sh
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5); b = torch.randn(cout) * 0.1
    y = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    for t in tiles_for(sh):
        rc = _capi.lib.m355_conv2d_fwd(P(x), B, H, W, cin, P(w), P(b), cout, 3, 1, 1, P(None), P(y), 0, t | (128 << 8),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        if rc == 0:
            ran.append((SHAPES.index(sh), t))      # a kernel that does not accept the shape returns an error and launches nothing
import json
json.dump(ran, open(OKFILE, "w"))
