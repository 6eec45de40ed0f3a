"""Generates tests/golden/dfine_golden.npz by calling the third-party functions the reference's D-FINE path uses
(transformers 5.15.0, models/d_fine/modeling_d_fine.py) on seeded inputs.  Run in the build container only
(`python tests/golden/make_dfine_golden.py`); the .npz is data: inputs and expected outputs."""
import os

import numpy as np
import torch
import transformers
from transformers.models.d_fine import modeling_d_fine as M
from transformers.models.d_fine.configuration_d_fine import DFineConfig

g = torch.Generator().manual_seed(0)
shapes = [(20, 20), (10, 10), (5, 5)]
S = sum(h * w for h, w in shapes)
out = {"transformers_version": np.array(transformers.__version__), "shapes": np.array(shapes, np.int32)}
B, Q, H, D = 2, 37, 8, 32
value = torch.randn(B, S, H, D, generator=g)
for tag, pts in (("a", [4, 4, 4]), ("b", [3, 6, 3])):
    P = sum(pts)
    # locations: most inside, some outside [0, 1] (zero padding), some exactly on the border
    loc = torch.rand(B, Q, H, P, 2, generator=g) * 1.3 - 0.15
    loc[0, 0, 0, 0] = torch.tensor([0.0, 0.0])
    loc[0, 0, 0, 1] = torch.tensor([1.0, 1.0])
    loc[0, 0, 0, 2] = torch.tensor([0.5, 0.975])
    attn = torch.softmax(torch.randn(B, Q, H, P, generator=g), -1)
    for method in ("default", "discrete"):
        y = M.multi_scale_deformable_attention_v2(value, shapes, loc, attn, pts, method)
        out[f"msda_{tag}_{method}"] = y.numpy()
    out[f"loc_{tag}"], out[f"attn_{tag}"], out[f"pts_{tag}"] = loc.numpy(), attn.numpy(), np.array(pts, np.int32)
out["value"] = value.numpy()
# the attention module itself (random init, 4-d reference points): inputs, the two linear layers' outputs, the result
torch.manual_seed(1)
cfgm = DFineConfig()
mod = M.DFineMultiscaleDeformableAttention(cfgm).eval()
hidden = torch.randn(B, Q, cfgm.d_model, generator=g)
enc = value.reshape(B, S, H * D)
refp = torch.rand(B, Q, 1, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4]) + torch.tensor([0.0, 0.0, 0.02, 0.02])
with torch.no_grad():
    ym, _ = mod(hidden, reference_points=refp, encoder_hidden_states=enc, spatial_shapes=torch.tensor(shapes),
                spatial_shapes_list=shapes)
    out["mod_offsets"] = mod.sampling_offsets(hidden).reshape(B, Q, H, 12, 2).numpy()
    out["mod_logits"] = mod.attention_weights(hidden).reshape(B, Q, H, 12).numpy()
out["mod_hidden"], out["mod_ref"], out["mod_out"] = hidden.numpy(), refp.reshape(B, Q, 4).numpy(), ym.numpy()
out["mod_w_off"], out["mod_b_off"] = mod.sampling_offsets.weight.detach().numpy(), mod.sampling_offsets.bias.detach().numpy()
out["mod_w_att"], out["mod_b_att"] = mod.attention_weights.weight.detach().numpy(), mod.attention_weights.bias.detach().numpy()
out["mod_offset_scale"] = np.array(mod.offset_scale, np.float32)
up, reg = torch.tensor([0.5]), torch.tensor([4.0])
proj = M.weighting_function(32, up, reg)
out["project"] = proj.numpy()
cfg = DFineConfig()
integ = M.DFineIntegral(cfg)
dist = torch.randn(3, 11, 4 * 33, generator=g) * 3
d = integ(dist, proj)
out["dist"], out["integral"] = dist.numpy(), d.numpy()
pts = torch.rand(3, 11, 4, generator=g)
pts[0, 0] = torch.tensor([3.0e38, -3.0e38, 5.0, 0.1])     # the reference feeds pre-sigmoid reference points
out["points"] = pts.numpy()
out["boxes"] = M.distance2bbox(pts, d, float(reg)).numpy()
out["boxes_clamped"] = M.distance2bbox(pts, d, float(reg)).clamp(0, 1).numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dfine_golden.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
