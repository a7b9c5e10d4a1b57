"""Oracle: feature statistics, normalisation, pairing, PCA / TICA / hTICA, linear projection.

TEST INFRASTRUCTURE (see oracle/__init__.py).  dtype follows the reference: float32 data,
float32 torch-CPU linear algebra for TICA, scikit-learn for PCA.
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------- statistics
def feature_stats(X: np.ndarray) -> dict:
    """mean / std(ddof=1) / min / max per column, as ``DataFrame.agg`` does on float32 columns.

    Follows modules/cv_learning/cv_calculator.py:294-297 (pandas agg; std is the sample std).
    """
    import pandas as pd

    df = pd.DataFrame(np.asarray(X, dtype=np.float32))
    stats_df = df.agg(["mean", "std", "min", "max"]).T
    return {s: stats_df[s].to_numpy() for s in ("mean", "std", "min", "max")}


def prepare_normalization(stats: dict, mode):
    """(mean, range) for the normalisation mode; |range| < 1e-8 -> 1.0.

    Follows cv_calculator.py:308-363.
    """
    if mode is None:
        means = np.zeros(len(stats["mean"]))
        ranges = np.ones(len(stats["mean"]))
    elif mode == "mean_std":
        means = stats["mean"]
        ranges = stats["std"]
    elif mode == "min_max_range1":
        means = stats["min"]
        ranges = stats["max"] - stats["min"]
    elif mode == "min_max_range2":
        means = (stats["min"] + stats["max"]) / 2
        ranges = (stats["max"] - stats["min"]) / 2
    else:
        raise ValueError(f"Normalization mode {mode} not recognized.")
    ranges = np.array(ranges, copy=True)
    ranges[np.abs(ranges) < 1e-8] = 1.0
    return np.asarray(means), ranges


def normalize(X: np.ndarray, mean, rng) -> np.ndarray:
    """float32 ``x.sub_(mean).div_(range)`` (cv_calculator.py:833-835); returns a new array."""
    t = torch.from_numpy(np.array(X, dtype=np.float32, copy=True))
    t.sub_(torch.tensor(np.asarray(mean), dtype=torch.float32))
    t.div_(torch.tensor(np.asarray(rng), dtype=torch.float32))
    return t.numpy()


# --------------------------------------------------------------------------- pairing
def timelagged_pairs(X, lag: int):
    """Pairs (i, i+lag), i = 0..N-lag-1, unit weights -- what mlcolvar's
    ``create_timelagged_dataset(X, lag_time=lag)`` yields for uniformly spaced frames
    (call sites cv_calculator.py:2247, 2309, 2544)."""
    X = torch.as_tensor(X)
    n = X.shape[0]
    return X[: n - lag], X[lag:]


# --------------------------------------------------------------------------- TICA
def correlation_matrix(x: torch.Tensor, y: torch.Tensor, symmetrize=True) -> torch.Tensor:
    """mlcolvar.core.stats.utils.correlation_matrix with unit weights (Appendix A.2)."""
    w = torch.ones(x.shape[0], dtype=x.dtype)
    corr = torch.einsum("ij, ik, i -> jk", x, y, w)
    corr = corr / torch.sum(w)
    if symmetrize:
        corr = 0.5 * (corr + corr.T)
    return corr


def cholesky_eigh(A: torch.Tensor, B: torch.Tensor, reg_B: float, n_eig: int = 0):
    """mlcolvar.core.stats.utils.cholesky_eigh (Appendix A.2): generalised eigenproblem
    A v = lambda (B + reg I) v, eigenvalues descending, unit-norm vectors, first row > 0."""
    B = B + reg_B * torch.eye(B.shape[0], dtype=B.dtype)
    L = torch.linalg.cholesky(B, upper=False)
    L_t = torch.t(L)
    L_i = torch.inverse(L)
    L_ti = torch.inverse(L_t)
    A_new = torch.matmul(torch.matmul(L_i, A), L_ti)
    eigvals, eigvecs = torch.linalg.eigh(A_new, UPLO="L")
    eigvals, indices = torch.sort(eigvals, 0, descending=True)
    eigvecs = eigvecs[:, indices]
    eigvecs = torch.matmul(L_ti, eigvecs)
    eigvecs = eigvecs / eigvecs.pow(2).sum(dim=0).sqrt()
    eigvecs = eigvecs * eigvecs[0, :].sign()
    if n_eig > 0:
        eigvals = eigvals[:n_eig]
        eigvecs = eigvecs[:, :n_eig]
    return eigvals, eigvecs


def tica_from_cov(C0: torch.Tensor, Ct: torch.Tensor, out: int, reg: float = 1e-6):
    """The eigen-solve half of TICA on already symmetrised covariances."""
    return cholesky_eigh(Ct, C0, reg, n_eig=min(out, C0.shape[0]))


def tica(x_t, x_lag, out: int, reg: float = 1e-6, dtype=torch.float32):
    """mlcolvar.core.stats.TICA.compute(data=[x_t, x_lag], remove_average=True)
    (call site cv_calculator.py:2257-2261; Appendix A.2).  Returns (evals, evecs, mean).
    If ``out`` exceeds the number of features all eigenvectors are returned (A.4)."""
    x_t = torch.as_tensor(x_t).to(dtype)
    x_lag = torch.as_tensor(x_lag).to(dtype)
    mu = x_t.mean(dim=0)  # mean of x_t only, removed from both
    xc = x_t - mu
    yc = x_lag - mu
    C0 = correlation_matrix(xc, xc)
    Ct = correlation_matrix(xc, yc)
    F = C0.shape[0]
    evals, evecs = cholesky_eigh(Ct, C0, reg, n_eig=min(out, F))
    return evals, evecs, mu


def tica_cv(Xn, lag: int, dim: int, dtype=torch.float32) -> np.ndarray:
    """TICACalculator.compute_cv (cv_calculator.py:2249-2267): weights F x dim."""
    x_t, x_lag = timelagged_pairs(Xn, lag)
    _, evecs, _ = tica(x_t, x_lag, dim, dtype=dtype)
    return evecs.numpy()


def htica_cv(Xn, lag: int, dim: int, num_subspaces: int, subspaces_dimension: int,
             dtype=torch.float32) -> np.ndarray:
    """HTICACalculator.compute_cv (cv_calculator.py:2311-2384; Appendix A.4)."""
    from scipy.sparse import block_diag

    x_t, x_lag = timelagged_pairs(Xn, lag)
    x_t = torch.as_tensor(x_t).to(dtype)
    x_lag = torch.as_tensor(x_lag).to(dtype)
    F = x_t.shape[1]
    split = F // num_subspaces
    if split == 0:
        raise ValueError("num_subspaces larger than number of features")
    level1, proj, proj_lag = [], [], []
    for a, b in zip(torch.split(x_t, split, dim=1), torch.split(x_lag, split, dim=1)):
        _, ev, _ = tica(a, b, subspaces_dimension, dtype=dtype)
        level1.append(ev.numpy())
        proj.append(a @ ev)
        proj_lag.append(b @ ev)
    T = block_diag(level1, format="csr")
    p = torch.cat(proj, dim=1)
    pl = torch.cat(proj_lag, dim=1)
    _, ev2, _ = tica(p, pl, dim, dtype=dtype)
    return np.asarray(T @ ev2.numpy())


# --------------------------------------------------------------------------- PCA
def pca_cv(Xn: np.ndarray, dim: int) -> np.ndarray:
    """PCACalculator.compute_cv (cv_calculator.py:2194-2215): sklearn PCA, components_.T,
    then flip each column so that row 0 is non-negative."""
    from sklearn.decomposition import PCA

    pca = PCA(n_components=dim)
    pca.fit(np.asarray(Xn))
    W = pca.components_.T.copy()
    for i in range(dim):
        if W[0, i] < 0:
            W[:, i] = -W[:, i]
    return W


# --------------------------------------------------------------------------- projection
def linear_cv_norm(Xn: np.ndarray, W: np.ndarray):
    """LinearCalculator.normalize_cv (cv_calculator.py:974-991): min/max of Xn @ W."""
    P = (torch.from_numpy(np.asarray(Xn, dtype=np.float32)) @ torch.tensor(W, dtype=torch.float32)).numpy()
    mn = P.min(axis=0)
    mx = P.max(axis=0)
    return (mx + mn) / 2, (mx - mn) / 2


def project_linear(X: np.ndarray, W, cv_mean, cv_range, feat_mean=None, feat_range=None) -> np.ndarray:
    """LinearCalculator.project_data (cv_calculator.py:918-972).  ``feat_mean`` given =>
    normalise first (normalize_data=True)."""
    if feat_mean is not None:
        X = normalize(X, feat_mean, feat_range)
    t = torch.from_numpy(np.asarray(X, dtype=np.float32)) @ torch.tensor(np.asarray(W), dtype=torch.float32)
    t = t.clone()
    t.sub_(torch.tensor(np.asarray(cv_mean), dtype=torch.float32))
    t.div_(torch.tensor(np.asarray(cv_range), dtype=torch.float32))
    return t.numpy()


def csv_round4(P: np.ndarray) -> np.ndarray:
    """The '%.4f' CSV seam (train_colvars_workflow.py:386 -> traj_cluster_workflow.py:202):
    float32 CVs are formatted with 4 decimals and re-read as float64."""
    P = np.asarray(P)
    flat = np.array([float("%.4f" % v) for v in P.ravel()], dtype=np.float64)
    return flat.reshape(P.shape)
