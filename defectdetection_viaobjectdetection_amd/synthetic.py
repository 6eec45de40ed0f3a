"""Synthetic B-scan workload: the device-resident batches `bench.py`, `smoke()` and the parity tests run on."""
import numpy as np
import torch


def synthetic_bscans(batch: int, h: int = 640, w: int = 640, seed: int = 0) -> np.ndarray:
    """uint8 (B,h,w,3), gray replicated; background clip(|N(0,35)|,0,255)
    (fixtures: mean ~27, max ~210) plus 1-3 bright horizontal bands 20-40 px tall."""
    g = torch.Generator().manual_seed(seed)
    bg = (torch.randn(batch, h, w, generator=g) * 35.0).abs().clamp_(0, 255)
    for b in range(batch):
        nb = int(torch.randint(1, 4, (1,), generator=g))
        for _ in range(nb):
            top = int(torch.randint(0, max(h - 40, 1), (1,), generator=g))
            tall = int(torch.randint(20, 41, (1,), generator=g))
            left = int(torch.randint(0, max(w // 2, 1), (1,), generator=g))
            wide = int(torch.randint(w // 8, w // 2 + 1, (1,), generator=g))
            amp = float(torch.randint(120, 211, (1,), generator=g))
            bg[b, top:top + tall, left:left + wide] = (bg[b, top:top + tall, left:left + wide] + amp).clamp_(0, 255)
    img = bg.round().to(torch.uint8).numpy()
    return np.repeat(img[..., None], 3, axis=3)
