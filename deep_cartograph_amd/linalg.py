"""Small dense host-side solves of the CV fit (F x F <= ~1024^2, not on the data-parallel path;
SURVEY.md section 8b): the generalised eigenproblem of TICA and the PCA eigen-decomposition,
in float64 on the covariance blocks produced (and all-reduced) by the HIP kernels."""
from __future__ import annotations

import contextlib
import os
from typing import Tuple

import numpy as np


def _few_threads(n: int = 1 << 30):
    """F x F solves with F of a few hundred are slower -- and on a 128-core host erratically so (tens of ms of
    thread start-up and contention) -- when the BLAS spreads them over every core: cap the pool for their duration.
    Not for tiny systems (the d x d TICA of a Deep-TICA batch record, d <= 16: no BLAS threads them, and building the
    limiter costs ~1 ms -- it was 70 % of a fit's host time when every loss record paid it)."""
    if n <= 64:
        return contextlib.nullcontext()
    try:
        from threadpoolctl import threadpool_limits

        return threadpool_limits(limits=min(8, os.cpu_count() or 1))
    except Exception:   # threadpoolctl missing: unrestricted, only slower
        return contextlib.nullcontext()


def tica_eigh(C0: np.ndarray, Ct: np.ndarray, reg: float = 1e-6, n_eig: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Ct v = lambda (C0 + reg I) v by the Cholesky reduction mlcolvar uses
    (mlcolvar.core.stats.utils.cholesky_eigh): eigenvalues descending,
    eigenvectors back-transformed, scaled to unit Euclidean norm, first row non-negative.
    n_eig > 0 keeps the leading n_eig pairs."""
    C0 = np.asarray(C0, dtype=np.float64)
    Ct = np.asarray(Ct, dtype=np.float64)
    n = C0.shape[0]
    with _few_threads(n):
        L = np.linalg.cholesky(C0 + reg * np.eye(n))
        Li = np.linalg.inv(L)
        A = Li @ Ct @ Li.T
        A = 0.5 * (A + A.T)
        evals, V = np.linalg.eigh(A)
        order = np.argsort(evals)[::-1]
        evals = evals[order]
        V = Li.T @ V[:, order]
    V = V / np.sqrt((V * V).sum(axis=0))
    sign = np.sign(V[0, :])
    sign[sign == 0] = 1.0
    V = V * sign
    if n_eig > 0:
        evals, V = evals[:n_eig], V[:, :n_eig]
    return evals, V


def tica_eigh_stack(C0: np.ndarray, Ct: np.ndarray, reg: float = 1e-6) -> Tuple[np.ndarray, np.ndarray]:
    """tica_eigh over a stack of small systems (C0, Ct: [m, d, d]) in one pass of numpy's stacked LAPACK drivers -- the same
    routine per matrix, the same numbers as m calls of tica_eigh: eigenvalues [m, d] descending, eigenvectors [m, d, d]."""
    C0 = np.asarray(C0, dtype=np.float64)
    Ct = np.asarray(Ct, dtype=np.float64)
    m, n = C0.shape[0], C0.shape[-1]
    if m == 0:
        return np.zeros((0, n)), np.zeros((0, n, n))
    with _few_threads(n):
        L = np.linalg.cholesky(C0 + reg * np.eye(n))
        Li = np.linalg.inv(L)
        LiT = np.swapaxes(Li, -1, -2)
        A = Li @ Ct @ LiT
        A = 0.5 * (A + np.swapaxes(A, -1, -2))
        evals, V = np.linalg.eigh(A)
        order = np.argsort(evals, axis=-1)[:, ::-1]
        evals = np.take_along_axis(evals, order, axis=-1)
        V = LiT @ np.take_along_axis(V, order[:, None, :], axis=-1)
    V = V / np.sqrt((V * V).sum(axis=-2, keepdims=True))
    sign = np.sign(V[:, :1, :])
    sign[sign == 0] = 1.0
    return evals, V * sign


def pca_components(C: np.ndarray, dim: int) -> np.ndarray:
    """Leading `dim` eigenvectors (F x dim) of the covariance, largest variance first, each
    column flipped so that its first entry is non-negative -- what sklearn PCA(n_components)
    .components_.T becomes after the reference's sign rule (cv_calculator.py:2204-2215)."""
    C = np.asarray(C, dtype=np.float64)
    with _few_threads(C.shape[0]):
        w, V = np.linalg.eigh(0.5 * (C + C.T))
    order = np.argsort(w)[::-1][:dim]
    W = V[:, order].copy()
    for i in range(W.shape[1]):
        if W[0, i] < 0:
            W[:, i] = -W[:, i]
    return W


def block_diag(blocks) -> np.ndarray:
    rows = sum(b.shape[0] for b in blocks)
    cols = sum(b.shape[1] for b in blocks)
    out = np.zeros((rows, cols), dtype=np.float64)
    r = c = 0
    for b in blocks:
        out[r:r + b.shape[0], c:c + b.shape[1]] = b
        r += b.shape[0]
        c += b.shape[1]
    return out
