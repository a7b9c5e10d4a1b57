"""Seeded synthetic feature matrices: k_slow independent AR(1) latent
chains mixed into F features + white noise + a per-column affine scramble.  Bench / test input
generation only -- torch is used as a random-number and elementwise utility here, none of this
is on the measured path."""
from __future__ import annotations

import math

import numpy as np
import torch

TIMESCALES = (2000.0, 700.0, 250.0, 90.0, 40.0, 20.0)
BASE_SEED = 20240607


def _mixing(F: int, k_slow: int, seed: int):
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.standard_normal((F, k_slow)) / math.sqrt(k_slow)
    scale = rng.uniform(0.1, 10.0, F)
    offset = rng.uniform(-5.0, 5.0, F)
    return A.astype(np.float32), scale.astype(np.float32), offset.astype(np.float32)


def synth_features(n: int, F: int, k_slow: int = 4, seed: int = BASE_SEED, shard: int = 0, device="cuda",
                   out: torch.Tensor | None = None, block: int = 4096, chunk: int = 1 << 20) -> torch.Tensor:
    """n x F float32 on ``device``.  The latent chains are exact AR(1) processes
    z[t+1] = rho z[t] + sqrt(1 - rho^2) eta, rho_j = exp(-1/T_j), generated block-wise;
    (seed, shard) selects an independent trajectory, the mixing matrix depends on seed only."""
    dev = torch.device(device)
    A, scale, offset = _mixing(F, k_slow, seed)
    A_t = torch.from_numpy(A).to(dev)
    scale_t = torch.from_numpy(scale).to(dev)
    offset_t = torch.from_numpy(offset).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed((seed * 1000003 + shard * 7919) % (2 ** 62))
    rho = torch.tensor([math.exp(-1.0 / TIMESCALES[j % len(TIMESCALES)]) for j in range(k_slow)], dtype=torch.float64, device=dev)
    sig = torch.sqrt(1.0 - rho * rho)
    t = torch.arange(block, dtype=torch.float64, device=dev)[:, None]
    pow_pos = rho[None, :] ** (t + 1.0)   # rho^(t+1)
    pow_t = rho[None, :] ** t             # rho^t
    pow_neg = rho[None, :] ** (-t)        # rho^(-t)
    X = out if out is not None else torch.empty(n, F, dtype=torch.float32, device=dev)
    z0 = torch.randn(k_slow, dtype=torch.float64, device=dev, generator=g)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        z = torch.empty(c1 - c0, k_slow, dtype=torch.float64, device=dev)
        for b0 in range(0, c1 - c0, block):
            b1 = min(c1 - c0, b0 + block)
            L = b1 - b0
            eta = torch.randn(L, k_slow, dtype=torch.float64, device=dev, generator=g)
            cs = torch.cumsum(sig[None, :] * eta * pow_neg[:L], dim=0)
            zb = pow_pos[:L] * z0[None, :] + pow_t[:L] * cs
            z[b0:b1] = zb
            z0 = zb[-1]
        eps = torch.randn(c1 - c0, F, dtype=torch.float32, device=dev, generator=g)
        xc = z.to(torch.float32) @ A_t.T
        xc.add_(eps, alpha=0.5)
        xc.mul_(scale_t).add_(offset_t)
        X[c0:c1] = xc
        del eps, xc, z
    return X
