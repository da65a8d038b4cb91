"""Synthetic MSM inputs resident in HBM: seeded scalars generated on the device with torch
(plumbing only) and the closed-form expected scalar for bases P_i = (t0 + i d) G."""
from __future__ import annotations

import torch


def random_scalars_device(n: int, r: int, seed: int, device: str = "cuda") -> torch.Tensor:
    """n x 8 little-endian 32-bit limbs (int32 storage), plain form, each value < r: the top limb is drawn
    below r's top limb, i.e. uniform on [0, top(r) * 2^224), which is [0, r) minus a 2^-29 relative tail."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    top = r >> 224
    lo = torch.randint(0, 1 << 32, (n, 7), generator=g, device=device, dtype=torch.int64)
    hi = torch.randint(0, top, (n, 1), generator=g, device=device, dtype=torch.int64)
    limbs = torch.cat([lo, hi], dim=1)
    # store as the uint32 bit patterns
    return (limbs & 0xFFFFFFFF).to(torch.uint32 if hasattr(torch, "uint32") else torch.int64).contiguous()


def closed_form_scalar(scalars: torch.Tensor, t0: int, d: int, r: int) -> int:
    """sum_i a_i (t0 + i d) mod r from the device tensor, with exact integer column sums."""
    n = scalars.shape[0]
    a = scalars.to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(n, device=a.device, dtype=torch.int64)
    ilo, ihi = idx & 0xFFF, idx >> 12
    s_a = 0
    s_ia = 0
    for k in range(8):
        col = a[:, k]
        s_a += int(col.sum().item()) << (32 * k)
        for h in range(2):
            half = (col >> (16 * h)) & 0xFFFF
            v = int((half * ilo).sum().item()) + (int((half * ihi).sum().item()) << 12)
            s_ia += v << (32 * k + 16 * h)
    return (t0 * s_a + d * s_ia) % r
