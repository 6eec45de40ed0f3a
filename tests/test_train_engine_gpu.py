"""Whole-network train-mode forward + backward on the HIP kernels vs PyTorch autograd through the CPU oracle
(SURVEY A13): same weights, same batch, BN with batch statistics, a fixed linear functional of the head outputs
as the loss so that d(loss)/d(outputs) is known exactly."""
import numpy as np
import pytest
import torch

from helpers import synthetic_bscans

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def _emulate_fp16_storage(oracle):
    """Round the oracle's conv outputs (z) and block outputs to fp16 in the forward pass and, through the same
    casts, their gradients in the backward pass: the precision at which the HIP path stores them."""
    import torch.nn as nn
    import yolov8_seg_oracle as orc

    def rnd(mod, inp, out):
        return out.half().float()
    for m in oracle.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, orc.Conv)) and m is not oracle.model[22].dfl.conv:
            m.register_forward_hook(rnd)


@pytest.mark.parametrize("emulate", [False, True])
@pytest.mark.parametrize("scale,shape,batch", [("n", (128, 160), 2), ("s", (96, 96), 3), ("n", (320, 320), 4)])
def test_train_forward_backward_parity(scale, shape, batch, emulate, cuda_device):
    import yolov8_seg_oracle as orc
    from defectdetection_viaobjectdetection_amd.spec import synthetic_state_dict
    from defectdetection_viaobjectdetection_amd.train_engine import TrainEngine
    nc = 1
    sd = synthetic_state_dict(scale, nc, seed=3)
    eng = TrainEngine(scale, nc, shape, batch)
    eng.load_state_dict(sd)
    oracle = orc.SegmentationModel(scale, nc)
    oracle.load_state_dict(sd)
    oracle.train()
    if emulate:
        _emulate_fp16_storage(oracle)
    imgs = synthetic_bscans(batch, shape[0], shape[1], seed=9)
    x = torch.from_numpy(imgs.transpose(0, 3, 1, 2).copy()).float() / 255.0
    raw_l, mc, protos = oracle.forward_raw(x)
    o_raw = torch.cat([r.view(batch, 64 + nc, -1) for r in raw_l], 2)
    o_raw = torch.cat((o_raw, mc), 1).permute(0, 2, 1)                     # (B, A, 97)
    g = torch.Generator().manual_seed(1)
    R1 = torch.randn(o_raw.shape, generator=g)
    R2 = torch.randn(protos.shape, generator=g)                            # (B,32,h,w)
    loss = (o_raw * R1).sum() + (protos * R2).sum()
    loss.backward()

    raw, pr = eng.forward(torch.from_numpy(imgs).to(cuda_device))
    torch.cuda.synchronize()
    e_raw = rel_l2(raw.cpu(), o_raw.detach())
    e_pr = rel_l2(pr.float().cpu().permute(0, 3, 1, 2), protos.detach())
    print(f"forward: raw rel-L2 {e_raw:.2e}  protos rel-L2 {e_pr:.2e}")
    # train-mode BN subtracts large post-SiLU channel means, which amplifies fp16 storage noise layer by layer
    # (tools/debug_train_fwd.py: 5e-4 after the stem -> ~2e-2 at the heads, smooth, no jump at any op)
    assert e_raw <= 4e-2 and e_pr <= 4e-2
    S = 1.0   # loss scale: with these weights the gradients grow to ~5e2 towards the stem, well inside fp16 range
    eng.backward(R1.to(cuda_device) * S, R2.permute(0, 2, 3, 1).contiguous().to(cuda_device) * S)
    torch.cuda.synchronize()
    osd = dict(oracle.named_parameters())
    worst, cos = [], []
    for name, p, gr in eng.trainable():
        ref = osd[name].grad
        got = gr.cpu() / S
        if got.dim() == 4 and not name.endswith("upsample.weight"):
            got = got.permute(0, 3, 1, 2)                                     # KRSC -> OIHW
        assert got.shape == ref.shape, name
        assert torch.isfinite(got).all(), name
        worst.append((rel_l2(got, ref), name))
        cos.append(float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0)))
    worst.sort(reverse=True)
    print("worst parameter-gradient rel-L2:", [(f"{e:.2e}", n) for e, n in worst[:6]])
    med = float(np.median([e for e, _ in worst]))
    print(f"median {med:.2e} over {len(worst)} parameter tensors; min cosine {min(cos):.4f}")
    # a wrong backward formula anywhere shows up as O(1) error / low cosine in every upstream tensor
    # (the float atomics of the BN statistics / split-K wgrad make the result vary by ~1e-2 in cosine from run to run)
    assert min(cos) >= 0.95 and worst[0][0] <= 0.32 and med <= 9e-2
    # running statistics follow torch's momentum update
    rm = eng.params["model.0.bn.running_mean"].cpu()
    assert torch.allclose(rm, oracle.model[0].bn.running_mean, atol=2e-3)
