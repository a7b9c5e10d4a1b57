"""Thin torch-tensor front end of the C-ABI (include/dcv.h).

torch is plumbing here: it owns device memory and the current stream; every number is
produced by a HIP kernel of libdcv.so.  All functions require CUDA(HIP) tensors and raise
otherwise -- there is no CPU path.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import DcvError, check


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DcvError("deep_cartograph_amd kernels need tensors on an MI355X (cuda) device; got a CPU tensor")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _check_matrix(X: torch.Tensor, dtype=torch.float32):
    if X.dim() != 2 or X.dtype != dtype or X.stride(1) != 1:
        raise DcvError(f"expected a row-major 2-D {dtype} tensor, got shape {tuple(X.shape)} dtype {X.dtype} strides {X.stride()}")


# ------------------------------------------------------------------------------- statistics
def col_stats_raw(X: torch.Tensor) -> torch.Tensor:
    """[4, F] float64: sum, sum of squares, min, max (dcv_col_stats)."""
    _require_gpu(X)
    _check_matrix(X)
    lib = _lib.load()
    n, F = X.shape
    out = torch.empty(4, F, dtype=torch.float64, device=X.device)
    ws = _ws(lib.dcv_col_stats_workspace(n, F), X.device)
    check(lib.dcv_col_stats(_ptr(X), n, F, X.stride(0), _ptr(out), _ptr(ws), ws.numel(), _stream()), "dcv_col_stats")
    return out


def finalize_stats(raw: torch.Tensor, n: int) -> dict:
    """mean / std(ddof=1) / min / max as float32 NumPy arrays from (all-reduced) raw sums."""
    r = raw.detach().cpu().numpy()
    mean = r[0] / n
    var = (r[1] - n * mean * mean) / max(n - 1, 1)
    std = np.sqrt(np.maximum(var, 0.0))
    return {"mean": mean.astype(np.float32), "std": std.astype(np.float32),
            "min": r[2].astype(np.float32), "max": r[3].astype(np.float32)}


def normalize(X: torch.Tensor, mean: torch.Tensor, rng: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(x - mean) / range in float32; ``out`` may be X itself (in place)."""
    _require_gpu(X, mean, rng, out)
    _check_matrix(X)
    if out is None:
        out = torch.empty_like(X)
    _check_matrix(out)
    n, F = X.shape
    check(_lib.load().dcv_normalize(_ptr(X), _ptr(out), n, F, X.stride(0), out.stride(0), _ptr(mean), _ptr(rng), _stream()),
          "dcv_normalize")
    return out


# ------------------------------------------------------------------------------- covariance
def lagged_cov_raw(X: torch.Tensor, n_pairs: int, lag: int, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Raw sums [a (F) | b (F) | A (F*F) | B (F*F)] float64 over pairs (i, i+lag), i < n_pairs."""
    _require_gpu(X, shift)
    _check_matrix(X)
    lib = _lib.load()
    n, F = X.shape
    if n_pairs + lag > n:
        raise DcvError(f"lagged_cov: n_pairs + lag = {n_pairs + lag} exceeds the {n} rows available")
    out = torch.empty(2 * F + 2 * F * F, dtype=torch.float64, device=X.device)
    ws = _ws(lib.dcv_lagged_cov_workspace(n_pairs, F, lag), X.device)
    check(lib.dcv_lagged_cov(_ptr(X), n_pairs, F, X.stride(0), lag, _ptr(shift), _ptr(out), _ptr(ws), ws.numel(), _stream()),
          "dcv_lagged_cov")
    return out


def covariances_from_raw(raw: np.ndarray, n_pairs: int, F: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(mu_z, C0, Ctau) in float64 from (all-reduced) raw sums, following
    mlcolvar TICA.compute: mean of x_t removed from both, both
    matrices divided by the number of pairs and symmetrised."""
    a = raw[:F]
    b = raw[F:2 * F]
    A = raw[2 * F:2 * F + F * F].reshape(F, F)
    B = raw[2 * F + F * F:].reshape(F, F)
    P = float(n_pairs)
    delta = a / P
    C0 = A / P - np.outer(delta, delta)
    Ct = B / P - np.outer(delta, b / P)
    C0 = 0.5 * (C0 + C0.T)
    Ct = 0.5 * (Ct + Ct.T)
    return delta, C0, Ct


# ------------------------------------------------------------------------------- projection
def project_linear(X: torch.Tensor, W: torch.Tensor, *, fmean=None, frange=None, bias=None, cvmean=None, cvrange=None,
                   want_out=True, want_minmax=False):
    """out = (((x - fmean)/frange) @ W + bias - cvmean)/cvrange ; optional per-column min/max."""
    _require_gpu(X, W, fmean, frange, bias, cvmean, cvrange)
    _check_matrix(X)
    lib = _lib.load()
    n, F = X.shape
    if W.dim() != 2 or W.shape[0] != F or not W.is_contiguous() or W.dtype != torch.float32:
        raise DcvError("project_linear: W must be a contiguous float32 F x d tensor")
    d = W.shape[1]
    out = torch.empty(n, d, dtype=torch.float32, device=X.device) if want_out else None
    mm = torch.empty(2, d, dtype=torch.float32, device=X.device) if want_minmax else None
    ws = _ws(lib.dcv_project_linear_workspace(n, F, d), X.device)
    check(lib.dcv_project_linear(_ptr(X), n, F, X.stride(0), _ptr(fmean), _ptr(frange), _ptr(W), d, _ptr(bias),
                                 _ptr(cvmean), _ptr(cvrange), _ptr(out), _ptr(mm), _ptr(ws), ws.numel(), _stream()),
          "dcv_project_linear")
    return out, mm


# ------------------------------------------------------------------------------- GEMM block
def gemm(mode: str, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Building block (tests): 'nt' A[M,K].B[N,K]^T, 'nn' A[M,K].B[K,N], 'tn' A[K,M]^T.B[K,N]."""
    _require_gpu(A, B)
    _check_matrix(A)
    _check_matrix(B)
    m = {"nt": 0, "nn": 1, "tn": 2}[mode]
    if m == 0:
        M, K = A.shape
        N = B.shape[0]
    elif m == 1:
        M, K = A.shape
        N = B.shape[1]
    else:
        K, M = A.shape
        N = B.shape[1]
    Cm = torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(_lib.load().dcv_gemm_f32(m, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(Cm), Cm.stride(0), M, N, K, _stream()),
          "dcv_gemm_f32")
    return Cm


def gemm_tn_split(A: torch.Tensor, B: torch.Tensor, k_chunk: int, slab_cap: int) -> torch.Tensor:
    """Split-K A[K,M]^T.B[K,N]: slabs [ceil(K / k_chunk), M, N]; raises when slab_cap slabs are too few."""
    _require_gpu(A, B)
    _check_matrix(A)
    _check_matrix(B)
    K, M = A.shape
    N = B.shape[1]
    slab = torch.full((slab_cap + 1, M, N), float("nan"), dtype=torch.float32, device=A.device)   # one guard slab behind the capacity
    check(_lib.load().dcv_gemm_tn_split(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(slab), slab_cap, M, N, K, int(k_chunk), _stream()),
          "dcv_gemm_tn_split")
    return slab


# ------------------------------------------------------------------------------- arithmetic mode
def set_gemm_mode(mode: str) -> None:
    """'split' (default): FP32-accurate split products on the BF16 matrix pipe; 'native': FP32-input MFMA."""
    lib = _lib.load()
    check(lib.dcv_set_gemm_mode({"native": 0, "split": 1}[mode]), "dcv_set_gemm_mode")


def get_gemm_mode() -> str:
    return "split" if _lib.load().dcv_get_gemm_mode() == 1 else "native"


# ------------------------------------------------------------------------------- k-means
def kmeans_step(P: torch.Tensor, centers: torch.Tensor, labels: torch.Tensor, offset: Optional[torch.Tensor] = None,
                want_mindist=False):
    """One Lloyd E-step + accumulation.  Returns (acc, mindist) with
    acc = [sums (k*d) | counts (k) | inertia | changed] float64 on the device."""
    _require_gpu(P, centers, labels, offset)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    n, d = P.shape
    k = centers.shape[0]
    acc = torch.empty(k * d + k + 2, dtype=torch.float64, device=P.device)
    md = torch.empty(n, dtype=torch.float64, device=P.device) if want_mindist else None
    ws = _ws(lib.dcv_kmeans_workspace(n, d, k), P.device)
    check(lib.dcv_kmeans_step(_ptr(P), n, d, _ptr(offset), _ptr(centers), k, _ptr(labels), _ptr(acc), _ptr(md), _ptr(ws), ws.numel(), _stream()),
          "dcv_kmeans_step")
    return acc, md


def kmeanspp_update(P: torch.Tensor, centre: torch.Tensor, closest: torch.Tensor, first: bool, offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """closest = first ? dist(., centre) : min(closest, dist(., centre)) in place; returns the new potential (1-element float64)."""
    _require_gpu(P, centre, closest, offset)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    n, d = P.shape
    if n == 0:   # an empty shard of a frame-sharded run contributes nothing
        return torch.zeros(1, dtype=torch.float64, device=P.device)
    pot = torch.empty(1, dtype=torch.float64, device=P.device)
    ws = _ws(lib.dcv_kmeanspp_workspace(n, 1), P.device)
    check(lib.dcv_kmeanspp_update(_ptr(P), n, d, _ptr(offset), _ptr(centre), 1 if first else 0, _ptr(closest), _ptr(pot), _ptr(ws), ws.numel(),
                                  _stream()), "dcv_kmeanspp_update")
    return pot


def kmeanspp_potentials(P: torch.Tensor, cand: torch.Tensor, closest: torch.Tensor, offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pot[t] = sum_i min(closest_i, dist(x_i, cand_t)) for the candidate centres cand (trials x d, offset frame)."""
    _require_gpu(P, cand, closest, offset)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    n, d = P.shape
    t = cand.shape[0]
    if n == 0:
        return torch.zeros(t, dtype=torch.float64, device=P.device)
    pot = torch.empty(t, dtype=torch.float64, device=P.device)
    ws = _ws(lib.dcv_kmeanspp_workspace(n, t), P.device)
    check(lib.dcv_kmeanspp_potentials(_ptr(P), n, d, _ptr(offset), _ptr(cand.contiguous()), t, _ptr(closest), _ptr(pot), _ptr(ws), ws.numel(),
                                      _stream()), "dcv_kmeanspp_potentials")
    return pot


def label_stats(P: torch.Tensor, labels: torch.Tensor, k: int, centers: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[sums (k*d) | counts (k) | sum ||x - c||^2 (k) | sum ||x - c|| (k)] per label, float64 on the device
    (the last two groups about `centers` when given)."""
    _require_gpu(P, labels, centers)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    n, d = P.shape
    acc = torch.empty(k * d + 3 * k, dtype=torch.float64, device=P.device)
    ws = _ws(lib.dcv_label_stats_workspace(n, d, k), P.device)
    check(lib.dcv_label_stats(_ptr(P), n, d, _ptr(labels), _ptr(centers), k, _ptr(acc), _ptr(ws), ws.numel(), _stream()), "dcv_label_stats")
    return acc


def cluster_dist_sums(Q: torch.Tensor, P_sorted: torch.Tensor, start: torch.Tensor) -> torch.Tensor:
    """S[i][c] = sum_j in cluster c ||Q_i - P_j||; P_sorted holds cluster c in rows [start[c], start[c+1])."""
    _require_gpu(Q, P_sorted, start)
    _check_matrix(Q, torch.float64)
    _check_matrix(P_sorted, torch.float64)
    lib = _lib.load()
    nq, d = Q.shape
    k = start.numel() - 1
    S = torch.empty(nq, k, dtype=torch.float64, device=Q.device)
    check(lib.dcv_cluster_dist_sums(_ptr(Q), nq, _ptr(P_sorted), _ptr(start), k, d, _ptr(S), _stream()), "dcv_cluster_dist_sums")
    return S


def silhouette_sum(S: torch.Tensor, qlabels: torch.Tensor, start: torch.Tensor) -> torch.Tensor:
    """Sum of the silhouette sample values of the queries (1-element float64 device tensor)."""
    _require_gpu(S, qlabels, start)
    lib = _lib.load()
    nq, k = S.shape
    out = torch.empty(1, dtype=torch.float64, device=S.device)
    ws = _ws(lib.dcv_silhouette_sum_workspace(nq), S.device)
    check(lib.dcv_silhouette_sum(_ptr(S), nq, k, _ptr(qlabels), _ptr(start), _ptr(out), _ptr(ws), ws.numel(), _stream()), "dcv_silhouette_sum")
    return out


def linear_binning(P: torch.Tensor, cols, lo, hi, bins: int):
    """Linear-binning weights of the points on a bins^d grid over [lo, hi] (d = len(cols) <= 2): (grid float64 device
    tensor of shape (bins,) or (bins, bins), number of points outside the bounds)."""
    import ctypes as C

    _require_gpu(P)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    d = len(cols)
    cols_a = (C.c_int32 * d)(*[int(c) for c in cols])
    lo_a = (C.c_double * d)(*[float(v) for v in lo])
    hi_a = (C.c_double * d)(*[float(v) for v in hi])
    grid = torch.empty((bins,) * d, dtype=torch.float64, device=P.device)
    ws = _ws(lib.dcv_linear_binning_workspace(d, bins), P.device)
    out = C.c_int64(0)
    check(lib.dcv_linear_binning(_ptr(P), P.shape[0], P.stride(0), d, cols_a, lo_a, hi_a, int(bins), _ptr(grid), C.byref(out), _ptr(ws),
                                 ws.numel(), _stream()), "dcv_linear_binning")
    return grid, int(out.value)


def nearest_rows(P: torch.Tensor, centers: torch.Tensor, row_offset: int = 0):
    """Per centroid: (distance, global row) of the nearest point (np.linalg.norm, first index on ties)."""
    _require_gpu(P, centers)
    _check_matrix(P, torch.float64)
    lib = _lib.load()
    n, d = P.shape
    k = centers.shape[0]
    dist = torch.empty(k, dtype=torch.float64, device=P.device)
    rows = torch.empty(k, dtype=torch.int64, device=P.device)
    ws = _ws(lib.dcv_nearest_rows_workspace(n, d, k), P.device)
    check(lib.dcv_nearest_rows(_ptr(P), n, d, _ptr(centers), k, row_offset, _ptr(dist), _ptr(rows), _ptr(ws), ws.numel(), _stream()),
          "dcv_nearest_rows")
    return dist, rows


def nearest_point(train: torch.Tensor, sup: torch.Tensor) -> torch.Tensor:
    _require_gpu(train, sup)
    _check_matrix(train, torch.float64)
    _check_matrix(sup, torch.float64)
    nn = torch.empty(sup.shape[0], dtype=torch.int64, device=sup.device)
    check(_lib.load().dcv_nearest_point(_ptr(train.contiguous()), train.shape[0], _ptr(sup.contiguous()), sup.shape[0],
                                        train.shape[1], _ptr(nn), _stream()), "dcv_nearest_point")
    return nn


# ------------------------------------------------------------------------------- MLP engine
class RcclComm:
    """RCCL communicator owned by the library (dcv_comm_*): the collectives of a data-parallel step are then issued by
    libdcv.so itself, in stream order with its kernels.  `dist` (an initialised torch.distributed, any backend) is used
    once, to hand rank 0's 128-byte unique id to the other ranks; world = 1 needs no bootstrap at all."""

    def __init__(self, dist=None, device="cuda"):
        import ctypes as C

        self.lib = _lib.load()
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            check(self.lib.dcv_comm_unique_id(uid), "dcv_comm_unique_id")
        if world > 1:
            backend = dist.get_backend()
            t = torch.tensor(list(uid), dtype=torch.uint8, device=device if backend == "nccl" else "cpu")
            dist.broadcast(t, src=0)
            uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        check(self.lib.dcv_comm_create(world, rank, uid, C.byref(h)), "dcv_comm_create")
        self.h, self.world, self.rank = h, world, rank

    def all_reduce(self, t: torch.Tensor, op: str = "sum"):
        _require_gpu(t)
        dt = {torch.float32: 0, torch.float64: 1}[t.dtype]
        check(self.lib.dcv_comm_allreduce(self.h, _ptr(t), t.numel(), dt, {"sum": 0, "min": 1, "max": 2}[op], _stream()), "dcv_comm_allreduce")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dcv_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mlp:
    """Owner of a ``dcv_mlp`` handle (Deep-TICA or autoencoder chain of Linear layers).

    ``dims`` = [F, ..., out]; ``acts`` one activation name per Linear.  Parameters are kept
    in the library's flat device buffer; ``set_linear`` / ``get_linear`` move torch-layout
    (out x in weight, out bias) tensors in and out."""

    def __init__(self, model: str, dims, acts, *, max_batch: int, lag: int = 0, tica_reg: float = 1e-6,
                 latent_layer: Optional[int] = None, lr: float = 1e-3, betas=(0.9, 0.999), eps: Optional[float] = None,
                 weight_decay: float = 0.0, optimizer: str = "Adam", amsgrad: bool = False, momentum: float = 0.0,
                 dampening: float = 0.0, nesterov: bool = False, alpha: float = 0.99, centered: bool = False,
                 lr_decay: float = 0.0, initial_accumulator_value: float = 0.0, opt_params=None, dropout=None, seed: int = 0,
                 batchnorm=None, bn_eps: float = 1e-5, bn_momentum: float = 0.1, maximize: bool = False, device="cuda"):
        import ctypes as C

        if not torch.cuda.is_available():
            raise DcvError("the MLP engine needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.model = model
        self.dims = [int(x) for x in dims]
        self.acts = list(acts)
        L = len(self.dims) - 1
        if len(self.acts) != L:
            raise DcvError("one activation per Linear layer expected")
        desc = _lib.MlpDesc()
        desc.model = {"deep_tica": _lib.MODEL_DEEPTICA, "ae": _lib.MODEL_AE}[model]
        desc.n_layers = L
        for i, v in enumerate(self.dims):
            desc.dims[i] = v
        for i, a in enumerate(self.acts):
            if a not in _lib.ACT:
                raise DcvError(f"activation {a!r} is not implemented by the HIP engine")
            desc.act[i] = _lib.ACT[a]
        desc.latent_layer = L if latent_layer is None else int(latent_layer)
        desc.lag = int(lag)
        desc.max_batch = int(max_batch)
        desc.tica_reg = float(tica_reg)
        if eps is None:   # torch's default for the optimiser
            eps = {"Adagrad": 1e-10, "Adadelta": 1e-6}.get(optimizer, 1e-8)
        desc.lr, desc.beta1, desc.beta2, desc.eps, desc.weight_decay = float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        if optimizer not in _lib.OPTIMIZER:
            raise DcvError(f"optimizer {optimizer!r} is not implemented by the HIP engine (have: {sorted(_lib.OPTIMIZER)})")
        desc.optimizer = _lib.OPTIMIZER[optimizer]
        desc.amsgrad, desc.nesterov, desc.centered = int(bool(amsgrad)), int(bool(nesterov)), int(bool(centered))
        desc.momentum, desc.dampening, desc.alpha = float(momentum), float(dampening), float(alpha)
        desc.lr_decay, desc.initial_accumulator_value = float(lr_decay), float(initial_accumulator_value)
        for i, v in enumerate(list(opt_params or [])[:4]):   # further constants of Adamax / NAdam / RAdam / Adadelta / ASGD / Rprop (dcv.h: DCV_OPT_*)
            desc.opt_p[i] = float(v)
        self.batchnorm = [bool(b) for b in (batchnorm if batchnorm is not None else [False] * L)]
        if len(self.batchnorm) != L:
            raise DcvError("one batchnorm flag per Linear layer expected")
        for i, b in enumerate(self.batchnorm):
            desc.batchnorm[i] = 1 if b else 0
        desc.bn_eps, desc.bn_momentum = float(bn_eps), float(bn_momentum)
        desc.maximize = 1 if maximize else 0
        self.dropout = [float(p or 0.0) for p in (dropout if dropout is not None else [0.0] * L)]
        if len(self.dropout) != L:
            raise DcvError("one dropout probability per Linear layer expected")
        for i, p in enumerate(self.dropout):
            desc.dropout[i] = p
        desc.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        h = C.c_void_p()
        check(self.lib.dcv_mlp_create(C.byref(desc), C.byref(h)), "dcv_mlp_create")
        self.h = h
        self.L = L
        self.latent_layer = desc.latent_layer
        self.max_batch = int(max_batch)
        self.rows_cap = 2 * self.max_batch if model == "deep_tica" else self.max_batch
        self.n_params = self.lib.dcv_mlp_num_params(self.h)
        self.offsets = [(self.lib.dcv_mlp_param_offset(self.h, l, 0), self.lib.dcv_mlp_param_offset(self.h, l, 1)) for l in range(L)]
        self.bn_offsets = [(self.lib.dcv_mlp_param_offset(self.h, l, 2), self.lib.dcv_mlp_param_offset(self.h, l, 3)) for l in range(L)]
        self.log_width = self.lib.dcv_mlp_log_width(self.h)
        self.stats_len = self.lib.dcv_mlp_stats_len(self.h)
        self._log_cap = 0
        self._dp = None            # state of data_parallel_step (callback thunk, tensor views of the library's buffers)
        self._upper_thunk = None   # ctypes thunk of backward(on_upper_grads=...)
        self._upper_fn = None
        self._upper_err = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.dcv_mlp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters
    def set_linears(self, linears, bn=None):
        """linears: list of (weight[out,in], bias[out]) CPU float32 tensors / arrays.  bn (optional): per Linear, None or a
        dict(weight, bias, running_mean, running_var, num_batches_tracked) of the BatchNorm1d behind it; layers with a
        normalisation and no entry start as a fresh BatchNorm1d (weight 1, bias 0, running mean 0 / variance 1)."""
        flat = np.zeros(self.n_params, dtype=np.float32)
        for l in range(self.L):
            if self.batchnorm[l]:
                go, bo_ = self.bn_offsets[l]
                d = (bn[l] if bn is not None else None) or {}
                flat[go: go + self.dims[l + 1]] = np.asarray(d.get("weight", np.ones(self.dims[l + 1])), dtype=np.float32)
                flat[bo_: bo_ + self.dims[l + 1]] = np.asarray(d.get("bias", np.zeros(self.dims[l + 1])), dtype=np.float32)
        for l, (w, b) in enumerate(linears):
            w = np.asarray(w, dtype=np.float32)
            b = np.asarray(b, dtype=np.float32)
            if w.shape != (self.dims[l + 1], self.dims[l]) or b.shape != (self.dims[l + 1],):
                raise DcvError(f"layer {l}: expected weight {(self.dims[l + 1], self.dims[l])}, got {w.shape}")
            wo, bo = self.offsets[l]
            flat[wo: wo + w.size] = w.ravel()
            flat[bo: bo + b.size] = b
        check(self.lib.dcv_mlp_set_params(self.h, flat.ctypes.data, _stream()), "dcv_mlp_set_params")
        if bn is not None:
            import ctypes as C

            for l in range(self.L):
                if self.batchnorm[l] and bn[l] is not None and "running_mean" in bn[l]:
                    rm = np.ascontiguousarray(np.asarray(bn[l]["running_mean"], dtype=np.float32))
                    rv = np.ascontiguousarray(np.asarray(bn[l]["running_var"], dtype=np.float32))
                    nbt = C.c_int64(int(bn[l].get("num_batches_tracked", 0)))
                    check(self.lib.dcv_mlp_bn_state(self.h, l, rm.ctypes.data, rv.ctypes.data, C.byref(nbt), 1, _stream()), "dcv_mlp_bn_state")

    def get_bn(self):
        """Per Linear: None, or the state of the BatchNorm1d behind it (weight, bias, running_mean, running_var, num_batches_tracked)."""
        import ctypes as C

        flat = np.empty(self.n_params, dtype=np.float32)
        check(self.lib.dcv_mlp_get_params(self.h, flat.ctypes.data, _stream()), "dcv_mlp_get_params")
        out = []
        for l in range(self.L):
            if not self.batchnorm[l]:
                out.append(None)
                continue
            o = self.dims[l + 1]
            go, bo_ = self.bn_offsets[l]
            rm, rv, nbt = np.empty(o, np.float32), np.empty(o, np.float32), C.c_int64(0)
            check(self.lib.dcv_mlp_bn_state(self.h, l, rm.ctypes.data, rv.ctypes.data, C.byref(nbt), 0, _stream()), "dcv_mlp_bn_state")
            out.append({"weight": flat[go: go + o].copy(), "bias": flat[bo_: bo_ + o].copy(), "running_mean": rm, "running_var": rv,
                        "num_batches_tracked": int(nbt.value)})
        return out

    def get_linears(self):
        flat = np.empty(self.n_params, dtype=np.float32)
        check(self.lib.dcv_mlp_get_params(self.h, flat.ctypes.data, _stream()), "dcv_mlp_get_params")
        out = []
        for l in range(self.L):
            wo, bo = self.offsets[l]
            o, i = self.dims[l + 1], self.dims[l]
            out.append((flat[wo: wo + o * i].reshape(o, i).copy(), flat[bo: bo + o].copy()))
        return out

    def _flat_view(self, ptr, n, dtype):
        # a torch view over library-owned device memory (for all-reduce / inspection)
        import ctypes as C

        class _Holder:
            pass

        itemsize = torch.tensor([], dtype=dtype).element_size()
        holder = _Holder()
        holder.__cuda_array_interface__ = {
            "shape": (n,), "typestr": {torch.float32: "<f4", torch.float64: "<f8"}[dtype],
            "data": (int(ptr), False), "version": 2, "strides": (itemsize,)}
        return torch.as_tensor(holder, device=self.device)

    def grads_view(self) -> torch.Tensor:
        return self._flat_view(self.lib.dcv_mlp_grads(self.h), self.n_params, torch.float32)

    def params_view(self) -> torch.Tensor:
        return self._flat_view(self.lib.dcv_mlp_params(self.h), self.n_params, torch.float32)

    def stats_view(self) -> torch.Tensor:
        return self._flat_view(self.lib.dcv_mlp_stats(self.h), self.stats_len, torch.float64)

    def set_lr(self, lr: float):
        check(self.lib.dcv_mlp_set_lr(self.h, float(lr)), "dcv_mlp_set_lr")

    def set_momentum(self, value: float):
        """beta1 (Adam family) / momentum (SGD, RMSprop) of the following updates (cycled by OneCycleLR)."""
        check(self.lib.dcv_mlp_set_momentum(self.h, float(value)), "dcv_mlp_set_momentum")

    def last_path(self) -> int:
        """0 = layer by layer, 1 = fused small-network autoencoder step, 2 = fused small-network Deep-TICA kernels."""
        return int(self.lib.dcv_mlp_last_path(self.h))

    def dropout_step(self) -> int:
        return int(self.lib.dcv_mlp_dropout_step(self.h))

    def dropout_mask(self, layer: int, step: int, rows: int) -> torch.Tensor:
        """keep / (1 - p) multipliers of the dropout behind Linear `layer` in training step `step` (test hook)."""
        out = torch.empty(rows, self.dims[layer + 1], dtype=torch.float32, device=self.device)
        check(self.lib.dcv_mlp_dropout_mask(self.h, int(layer), int(step), int(rows), _ptr(out), _stream()), "dcv_mlp_dropout_mask")
        return out

    def set_row_sharing(self, enable: bool):
        """Deep-TICA contiguous batches: share the rows of the two halves (default) or not."""
        check(self.lib.dcv_mlp_set_row_sharing(self.h, 1 if enable else 0), "dcv_mlp_set_row_sharing")

    def set_feature_range(self, rng):
        r = np.ascontiguousarray(np.asarray(rng, dtype=np.float32))
        check(self.lib.dcv_mlp_set_feature_range(self.h, r.ctypes.data, _stream()), "dcv_mlp_set_feature_range")

    # -- steps
    def _args(self, Xn, idx, row0, batch):
        _require_gpu(Xn, idx)
        _check_matrix(Xn)
        if idx is not None and (idx.dtype != torch.int64 or not idx.is_contiguous()):
            raise DcvError("batch indices must be a contiguous int64 device tensor")
        return _ptr(Xn), Xn.stride(0), _ptr(idx), int(row0), int(batch)

    def forward(self, Xn, idx=None, row0=0, batch=None, train=True):
        batch = int(batch if batch is not None else idx.numel())
        check(self.lib.dcv_mlp_forward(self.h, *self._args(Xn, idx, row0, batch), 1 if train else 0, _stream()), "dcv_mlp_forward")

    def backward(self, Xn, idx=None, row0=0, batch=None, global_batch=None, train=True, on_upper_grads=None):
        """`on_upper_grads`: host callable invoked inside the call once the gradients of layers 1.. are reduced into
        upper_grads_view() and before the layer-0 weight gradient is enqueued (data-parallel overlap hook)."""
        import ctypes as C

        batch = int(batch if batch is not None else idx.numel())
        gb = int(global_batch if global_batch is not None else batch)
        use_cb = on_upper_grads is not None and train and self.L > 1
        if use_cb:
            if self._upper_thunk is None:   # one thunk per engine; the Python callable of the current call sits in a cell

                def _tramp(_u):
                    try:
                        self._upper_fn()
                    except BaseException as e:   # ctypes would print and swallow it
                        self._upper_err = e

                self._upper_thunk = C.CFUNCTYPE(None, C.c_void_p)(_tramp)
            self._upper_fn, self._upper_err = on_upper_grads, None
            check(self.lib.dcv_mlp_set_upper_grads_callback(self.h, C.cast(self._upper_thunk, C.c_void_p), None), "dcv_mlp_set_upper_grads_callback")
        try:
            check(self.lib.dcv_mlp_backward(self.h, *self._args(Xn, idx, row0, batch), gb, 1 if train else 0, _stream()), "dcv_mlp_backward")
        finally:
            if use_cb:
                check(self.lib.dcv_mlp_set_upper_grads_callback(self.h, None, None), "dcv_mlp_set_upper_grads_callback")
                self._upper_fn = None
        if use_cb and self._upper_err is not None:
            e, self._upper_err = self._upper_err, None
            raise e

    def upper_grads_view(self) -> torch.Tensor:
        """Gradients of layers 1.. (the tail of the flat buffer)."""
        return self.grads_view()[self.offsets[1][0]:] if self.L > 1 else self.grads_view()[:0]

    def layer0_grads_view(self) -> torch.Tensor:
        return self.grads_view()[: self.offsets[1][0]] if self.L > 1 else self.grads_view()

    def data_parallel_step(self, Xn, dist, global_batch, idx=None, row0=0, batch=None, train=True, group=None):
        """One synchronous data-parallel step over `dist` (torch.distributed) as ONE library call (dcv_mlp_dp_step):
        forward, all-reduce of the batch statistics, backward, all-reduce of the gradient buffer (from 32 768 rows per
        rank up in two pieces, the upper layers' started under the layer-0 weight gradient; DCV_DP_OVERLAP=0|1 forces
        either form), optimiser update.  The collectives run in a host callback; an exception raised there (an RCCL
        error, say) aborts the step inside the library and is re-raised here -- never swallowed."""
        import ctypes as C

        rows = int(batch if batch is not None else idx.numel())
        if isinstance(dist, RcclComm):   # the library's own communicator: nothing of the step runs in Python
            env = os.environ.get("DCV_DP_OVERLAP")
            overlap = (rows >= 32768) if env is None else env == "1"
            check(self.lib.dcv_comm_bind_stream(dist.h, _stream()), "dcv_comm_bind_stream")
            check(self.lib.dcv_mlp_dp_step(self.h, *self._args(Xn, idx, row0, rows), int(global_batch), 1 if train else 0, 1 if overlap else 0,
                                           self.lib.dcv_comm_dp_allreduce_fn(), dist.h, _stream()), "dcv_mlp_dp_step")
            return
        env = os.environ.get("DCV_DP_OVERLAP")
        # The two-piece form pays one more collective launch (~15-20 us of latency at these message sizes) to hide the
        # upper layers' all-reduce under the layer-0 weight gradient: worth it only when that product outlasts a
        # collective -- not at a few thousand rows per rank (8192-pair global batch over 1-8 ranks).
        overlap = (rows >= 32768) if env is None else env == "1"
        st = self._dp
        if st is None:
            st = self._dp = {"views": {}, "pending": [], "error": None, "dist": None, "group": None}

            def _cb(_user, buf, count, dtype, phase):
                try:
                    if phase == 3:   # DCV_DP_WAIT
                        for w in st["pending"]:
                            w.wait()
                        st["pending"].clear()
                        return 0
                    key = (int(buf), int(count), int(dtype))
                    t = st["views"].get(key)
                    if t is None:
                        t = st["views"][key] = self._flat_view(buf, int(count), torch.float64 if dtype == 1 else torch.float32)
                    d = st["dist"]
                    if phase == 2:   # DCV_DP_UPPER_START
                        st["pending"].append(d.all_reduce(t, op=d.ReduceOp.SUM, group=st["group"], async_op=True))
                    else:
                        d.all_reduce(t, op=d.ReduceOp.SUM, group=st["group"])
                    return 0
                except BaseException as e:   # ctypes would print and swallow it: stash, fail the step, re-raise outside
                    st["error"] = e
                    return 1

            st["thunk"] = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)(_cb)
        st["dist"], st["group"], st["error"] = dist, group, None
        st["pending"].clear()
        rc = self.lib.dcv_mlp_dp_step(self.h, *self._args(Xn, idx, row0, rows), int(global_batch), 1 if train else 0, 1 if overlap else 0,
                                      C.cast(st["thunk"], C.c_void_p), None, _stream())
        if st["error"] is not None:
            err, st["error"] = st["error"], None
            raise err
        check(rc, "dcv_mlp_dp_step")

    def set_rank(self, rank: int):
        """Rank of this engine in a data-parallel run (independent dropout masks per rank)."""
        check(self.lib.dcv_mlp_set_rank(self.h, int(rank)), "dcv_mlp_set_rank")

    def layer_output(self, layer: int, rows: int) -> torch.Tensor:
        """Post-activation output of Linear `layer` in the last forward (test hook)."""
        out = torch.empty(int(rows), self.dims[layer + 1], dtype=torch.float32, device=self.device)
        check(self.lib.dcv_mlp_layer_output(self.h, int(layer), int(rows), _ptr(out), _stream()), "dcv_mlp_layer_output")
        return out

    def apply(self):
        check(self.lib.dcv_mlp_apply(self.h, _stream()), "dcv_mlp_apply")

    def set_graph(self, enable: bool):
        """Opt in to hipGraph launches of the step entry points (needs a non-null stream)."""
        check(self.lib.dcv_mlp_set_graph(self.h, 1 if enable else 0), "dcv_mlp_set_graph")

    def graph_launches(self) -> int:
        """Calls of train_step / forward / backward / eval_step that went out as one hipGraph launch."""
        return int(self.lib.dcv_mlp_graph_launches(self.h))

    def train_step(self, Xn, idx=None, row0=0, batch=None):
        batch = int(batch if batch is not None else idx.numel())
        check(self.lib.dcv_mlp_train_step(self.h, *self._args(Xn, idx, row0, batch), _stream()), "dcv_mlp_train_step")

    def eval_step(self, Xn, idx=None, row0=0, batch=None):
        batch = int(batch if batch is not None else idx.numel())
        check(self.lib.dcv_mlp_eval_step(self.h, *self._args(Xn, idx, row0, batch), _stream()), "dcv_mlp_eval_step")

    def train_steps(self, Xn, batch: int, nsteps: int, idx=None, row0=0):
        """`nsteps` training steps in one call (constant learning rate): step j on idx[j * batch:(j + 1) * batch] (or rows
        row0 + j * batch ...), identical to nsteps train_step calls -- the per-call cost of the host is paid once."""
        batch, nsteps = int(batch), int(nsteps)
        if idx is not None and idx.numel() < batch * nsteps:
            raise DcvError(f"train_steps: {idx.numel()} indices for {nsteps} batches of {batch}")
        if idx is None and int(row0) + batch * nsteps > Xn.shape[0]:
            raise DcvError(f"train_steps: rows {row0} + {nsteps} x {batch} exceed the matrix ({Xn.shape[0]} rows)")
        check(self.lib.dcv_mlp_train_steps(self.h, *self._args(Xn, idx, row0, batch), nsteps, _stream()), "dcv_mlp_train_steps")

    def eval_steps(self, Xn, batch: int, nbatches: int, idx=None, row0=0):
        """A validation pass: `nbatches` evaluation steps of `batch` samples, batch j = idx[j * batch:(j + 1) * batch] (or rows
        row0 + j * batch ...), one loss record each in batch order -- the records `nbatches` eval_step calls would append.
        Small networks run many batches per launch (dcv_mlp_eval_steps, include/dcv.h)."""
        batch, nbatches = int(batch), int(nbatches)
        if idx is not None and idx.numel() < batch * nbatches:
            raise DcvError(f"eval_steps: {idx.numel()} indices for {nbatches} batches of {batch}")
        if idx is None and int(row0) + batch * nbatches > Xn.shape[0]:
            raise DcvError(f"eval_steps: rows {row0} + {nbatches} x {batch} exceed the matrix ({Xn.shape[0]} rows)")
        check(self.lib.dcv_mlp_eval_steps(self.h, *self._args(Xn, idx, row0, batch), nbatches, _stream()), "dcv_mlp_eval_steps")

    # -- metrics log
    def reset_log(self, capacity: int):
        check(self.lib.dcv_mlp_reset_log(self.h, int(capacity), _stream()), "dcv_mlp_reset_log")
        self._log_cap = max(self._log_cap, int(capacity))

    def read_log(self) -> np.ndarray:
        import ctypes as C

        out = np.empty((max(self._log_cap, 1), self.log_width), dtype=np.float64)
        n = C.c_int32(0)
        check(self.lib.dcv_mlp_read_log(self.h, out.ctypes.data, out.shape[0], C.byref(n), _stream()), "dcv_mlp_read_log")
        return out[: n.value].copy()

    # -- per-kernel timing (HIP events on the launch stream)
    def profile_begin(self, max_steps: int, level: int = 1):
        check(self.lib.dcv_mlp_profile_begin(self.h, int(max_steps), int(level)), "dcv_mlp_profile_begin")

    def profile_pause(self, paused: bool, skip_kinds=()):
        """Steps issued while paused carry no events (a timed region can be sampled); skip_kinds: launch kinds
        ('fwd', 'wgrad', 'dgrad') left unsampled while not paused."""
        code = 1 if paused else 0
        for k in skip_kinds:
            code |= 2 << ("fwd", "wgrad", "dgrad").index(k)
        check(self.lib.dcv_mlp_profile_pause(self.h, code), "dcv_mlp_profile_pause")

    def profile_end(self):
        """{(layer, kind): (total_ms, launches)} with kind in 'fwd' | 'wgrad' | 'dgrad'."""
        ms = np.zeros(3 * self.L, dtype=np.float64)
        cnt = np.zeros(3 * self.L, dtype=np.int32)
        check(self.lib.dcv_mlp_profile_end(self.h, ms.ctypes.data, cnt.ctypes.data), "dcv_mlp_profile_end")
        kinds = ("fwd", "wgrad", "dgrad")
        return {(c // 3, kinds[c % 3]): (float(ms[c]), int(cnt[c])) for c in range(3 * self.L) if cnt[c] > 0}

    # -- inference
    def input_sensitivity(self, Xn, gout, scale):
        """sum over the rows of Xn of |d(sum_j cv_j)/d xn_i| * scale_i  (float64 [F], device),
        chunked to the engine's row capacity.  gout [d_latent]: gradient of the summed CV with
        respect to the network output; scale [F]."""
        _require_gpu(Xn, gout, scale)
        _check_matrix(Xn)
        n, F = Xn.shape
        total = torch.zeros(F, dtype=torch.float64, device=Xn.device)
        part = torch.empty(F, dtype=torch.float64, device=Xn.device)
        cap = min(self.rows_cap, max(1, n))
        nbytes = self.lib.dcv_mlp_input_sensitivity_workspace(self.h, cap)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=Xn.device)
        gout = gout.to(torch.float32).contiguous()
        scale = scale.to(torch.float32).contiguous()
        for s in range(0, n, cap):
            e = min(n, s + cap)
            check(self.lib.dcv_mlp_input_sensitivity(self.h, _ptr(Xn[s:e]), e - s, Xn.stride(0), _ptr(gout), _ptr(scale), _ptr(part),
                                                     _ptr(ws), ws.numel(), _stream()), "dcv_mlp_input_sensitivity")
            total += part
        return total

    def infer(self, Xn, *, tmean=None, tevecs=None, pmean=None, prange=None, want_out=True, want_minmax=False):
        """Forward of every row of Xn (chunked to the engine's row capacity).
        Returns (out [n, d] or None, minmax [2, d] or None)."""
        _require_gpu(Xn, tmean, tevecs, pmean, prange)
        _check_matrix(Xn)
        n = Xn.shape[0]
        d = self.dims[self.latent_layer]
        out = torch.empty(n, d, dtype=torch.float32, device=Xn.device) if want_out else None
        mm = None
        for s in range(0, n, self.rows_cap):
            e = min(n, s + self.rows_cap)
            chunk = Xn[s:e]
            mmc = torch.empty(2, d, dtype=torch.float32, device=Xn.device) if want_minmax else None
            check(self.lib.dcv_mlp_infer(self.h, _ptr(chunk), e - s, Xn.stride(0), _ptr(tmean), _ptr(tevecs), _ptr(pmean), _ptr(prange),
                                         _ptr(out[s:e]) if want_out else None, _ptr(mmc), _stream()), "dcv_mlp_infer")
            if want_minmax:
                if mm is None:
                    mm = mmc
                else:  # merging 2*d extrema of chunks
                    mm = torch.stack([torch.minimum(mm[0], mmc[0]), torch.maximum(mm[1], mmc[1])])
        return out, mm
