"""Time the D-FINE deformable-attention core at BASELINE config 5's shape (batch 16 of 640x640: 300 queries, 8 heads,
80^2 + 40^2 + 20^2 value map, 3 x 4 points): the HIP kernel through the C-ABI, the same transformers function run by
PyTorch-ROCm on the GPU (grid_sample + permutes), and the numpy oracle on the host (one batch element)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from defectdetection_viaobjectdetection_amd import dfine
g = torch.Generator().manual_seed(0)
shapes = [(80, 80), (40, 40), (20, 20)]; pts = [4, 4, 4]
B, S, Q, H, D, P = 16, 8400, 300, 8, 32, 12
value = torch.randn(B, S, H, D, generator=g).cuda()
loc = (torch.rand(B, Q, H, P, 2, generator=g) * 1.1 - 0.05).cuda()
attn = torch.softmax(torch.randn(B, Q, H, P, generator=g), -1).cuda()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
us = timeit(lambda: dfine.multi_scale_deformable_attention_v2(value, shapes, loc, attn, pts))
gather = B * Q * H * P * 4 * D * 4           # four 128-byte corner reads per sampling point
unique = value.numel() * 4 + loc.numel() * 4 + attn.numel() * 4 + B * Q * H * D * 4
out = {"op": "multi_scale_deformable_attention_v2", "shape": dict(B=B, S=S, Q=Q, heads=H, head_dim=D, points=pts),
       "hip_us": round(us, 1), "gather_GBps": round(gather / us / 1e3, 1), "unique_bytes_GBps": round(unique / us / 1e3, 1),
       "gather_MB": round(gather / 1e6, 1), "unique_MB": round(unique / 1e6, 1)}
try:
    from transformers.models.d_fine import modeling_d_fine as M
    t_us = timeit(lambda: M.multi_scale_deformable_attention_v2(value, shapes, loc, attn, pts, "default"), 20)
    ref = M.multi_scale_deformable_attention_v2(value, shapes, loc, attn, pts, "default")
    mine = dfine.multi_scale_deformable_attention_v2(value, shapes, loc, attn, pts)
    out["torch_rocm_us"] = round(t_us, 1)
    out["max_abs_diff_vs_torch_rocm"] = float((ref - mine).abs().max())
except Exception as ex:  # transformers missing on the box
    out["torch_rocm_us"] = None; out["note"] = repr(ex)
# the whole attention module (two linear layers + softmax + locations + core), random-init weights
try:
    from transformers.models.d_fine.configuration_d_fine import DFineConfig
    mod = M.DFineMultiscaleDeformableAttention(DFineConfig()).cuda().eval()
    hidden = torch.randn(B, Q, 256, generator=g).cuda()
    refp = (torch.rand(B, Q, 1, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4]) + torch.tensor([0.0, 0.0, 0.02, 0.02])).cuda()
    enc = value.reshape(B, S, H * D)
    with torch.no_grad():
        f_ref = lambda: mod(hidden, reference_points=refp, encoder_hidden_states=enc, spatial_shapes=torch.tensor(shapes), spatial_shapes_list=shapes)[0]  # noqa: E731
        f_hip = lambda: dfine.deformable_attention(hidden, refp, enc, shapes, mod.sampling_offsets, mod.attention_weights, pts, 8, mod.offset_scale)  # noqa: E731
        out["module_torch_rocm_us"] = round(timeit(f_ref, 20), 1)
        out["module_hip_us"] = round(timeit(f_hip, 50), 1)
        out["module_max_abs_diff"] = float((f_ref() - f_hip()).abs().max())
except Exception as ex:
    out["module_note"] = repr(ex)
import dfine_oracle as orc
v1, l1, a1 = value[:1].cpu().numpy(), loc[:1].cpu().numpy(), attn[:1].cpu().numpy()
t0 = time.perf_counter(); orc.multi_scale_deformable_attention_v2(v1, shapes, l1, a1, pts); el = time.perf_counter() - t0
out["cpu_oracle_us_per_batch16"] = round(el * 16 * 1e6, 0)
# decode
proj = dfine.weighting_function(32, torch.tensor([0.5]).cuda(), 4.0)
dist = torch.randn(B, Q, 4 * 33, generator=g).cuda(); ref_pts = torch.rand(B, Q, 4, generator=g).cuda()
out["decode_us"] = round(timeit(lambda: dfine.decode_boxes(dist, proj, ref_pts, 4.0, True)), 1)
print(json.dumps(out))
