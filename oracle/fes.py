"""Oracle: free-energy surface from a binned Gaussian kernel density estimate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates what figures.plot_fes asks of
mlcolvar.utils.fes.compute_fes(..., backend="KDEpy") (figures.py:95-98): KDEpy FFTKDE = linear binning of the samples on
the evaluation grid, convolution of the binned weights with the kernel, then F = -kB T log(density + eps) with a zero
minimum.  mlcolvar==1.2.2 and KDEpy are third-party packages absent from /root/reference and from this image: restated
from their published algorithms, PARITY UNPINNED (no reference fixture holds an FES); `exact_kde_fes` is the direct
O(n * nodes) Gaussian sum the binned form approximates."""
from __future__ import annotations

import numpy as np

KB_KJ_MOL = 0.0083144621


def linear_binning(X: np.ndarray, lo, hi, bins: int) -> np.ndarray:
    """Unit weight of every point spread over its 2^d neighbouring nodes (d = 1, 2); points outside [lo, hi] are ignored."""
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    t = [(X[:, c] - lo[c]) * ((bins - 1) / (hi[c] - lo[c])) for c in range(d)]
    ok = np.ones(n, dtype=bool)
    for c in range(d):
        ok &= (t[c] >= 0) & (t[c] <= bins - 1)
    idx, frac = [], []
    for c in range(d):
        i0 = np.minimum(t[c][ok].astype(np.int64), bins - 2)
        idx.append(i0)
        frac.append(t[c][ok] - i0)
    grid = np.zeros((bins,) * d)
    if d == 1:
        np.add.at(grid, idx[0], 1.0 - frac[0])
        np.add.at(grid, idx[0] + 1, frac[0])
    else:
        f0, f1 = frac
        np.add.at(grid, (idx[0], idx[1]), (1 - f0) * (1 - f1))
        np.add.at(grid, (idx[0], idx[1] + 1), (1 - f0) * f1)
        np.add.at(grid, (idx[0] + 1, idx[1]), f0 * (1 - f1))
        np.add.at(grid, (idx[0] + 1, idx[1] + 1), f0 * f1)
    return grid


def binned_fes(X: np.ndarray, temperature: float, bandwidth: float, bins: int, lo, hi, eps: float = 1e-10) -> np.ndarray:
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    d = X.shape[1]
    dens = linear_binning(X, lo, hi, bins) / X.shape[0]
    for c in range(d):
        h = (hi[c] - lo[c]) / (bins - 1)
        half = int(np.ceil(6.0 * bandwidth / h))
        t = np.arange(-half, half + 1) * h
        k = np.exp(-0.5 * (t / bandwidth) ** 2) / (bandwidth * np.sqrt(2.0 * np.pi))
        dens = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), c, dens)
    fes = -KB_KJ_MOL * temperature * np.log(dens + eps)
    fes -= fes.min()
    return fes.T if d == 2 else fes


def exact_kde_fes(X: np.ndarray, temperature: float, bandwidth: float, bins: int, lo, hi, eps: float = 1e-10) -> np.ndarray:
    """Direct Gaussian KDE on the same grid (small n only)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    n, d = X.shape
    axes = [np.linspace(lo[c], hi[c], bins) for c in range(d)]
    dens = np.ones((bins,) * d)
    if d == 1:
        dens = np.exp(-0.5 * ((axes[0][:, None] - X[None, :, 0]) / bandwidth) ** 2).sum(1) / (n * bandwidth * np.sqrt(2 * np.pi))
    else:
        k0 = np.exp(-0.5 * ((axes[0][:, None] - X[None, :, 0]) / bandwidth) ** 2)   # [bins, n]
        k1 = np.exp(-0.5 * ((axes[1][:, None] - X[None, :, 1]) / bandwidth) ** 2)
        dens = (k0 @ k1.T) / (n * bandwidth ** 2 * 2 * np.pi)
    fes = -KB_KJ_MOL * temperature * np.log(dens + eps)
    fes -= fes.min()
    return fes.T if d == 2 else fes
