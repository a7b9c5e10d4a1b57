"""Clustering of the projected trajectories: mirror of the reference's
deep_cartograph/modules/statistics/statistics.py (optimize_clustering :17-110, cluster_data
:112-157, kmeans_clustering :159-197, find_centroids :337-379).

k-means runs on the GPU: the Lloyd iterations (assignment, per-cluster sums, inertia, changed
labels) are one HIP pass each over the float64 points, k-means++ seeding and the convergence
logic follow scikit-learn's KMeans(random_state=0) step by step on the host (SURVEY.md Appendix
A.8), so the labels equal the reference's.  Hierarchical clustering and HDBSCAN keep delegating
to scikit-learn, as the scope table says (O(N^2) tree algorithms, out of scope)."""
from __future__ import annotations

import logging
import sys
from typing import Dict, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from . import hip
from .parallel import Comm, global_topk, reduce_nearest

logger = logging.getLogger(__name__)

KMEANS_TOL = 1e-4       # sklearn default
KMEANS_MAX_ITER = 300   # sklearn default


def _device():
    if not torch.cuda.is_available():
        raise hip.DcvError("deep_cartograph_amd needs an MI355X for k-means: no GPU visible and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class _DevicePoints:
    """Points resident on the GPU (float64, this rank's shard) with their global statistics."""

    def __init__(self, features: np.ndarray, comm: Comm):
        self.comm = comm
        self.dev = _device()
        self.P = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float64)).to(self.dev)
        self.n_local, self.d = self.P.shape
        # mean and mean(var) of the data through the k-means kernel itself (k = 1, centre 0):
        # sums -> mean, inertia -> sum ||x||^2
        zero = torch.zeros(1, self.d, dtype=torch.float64, device=self.dev)
        lab = torch.zeros(self.n_local, dtype=torch.int32, device=self.dev)
        acc, _ = hip.kmeans_step(self.P, zero, lab)
        acc = comm.sum_(acc).cpu().numpy()
        self.n = int(round(acc[self.d]))
        self.mean = acc[: self.d] / self.n
        self.tol_abs = (acc[self.d + 1] / self.n - float((self.mean ** 2).sum())) / self.d * KMEANS_TOL
        self.mean_t = torch.from_numpy(self.mean).to(self.dev)
        self.labels = torch.full((self.n_local,), -1, dtype=torch.int32, device=self.dev)

    def step(self, centers_c: np.ndarray, want_mindist=False):
        """One E-step + accumulation against centred centres; returns host (sums, counts, inertia, changed)."""
        k = centers_c.shape[0]
        c = torch.from_numpy(np.ascontiguousarray(centers_c)).to(self.dev)
        acc, md = hip.kmeans_step(self.P, c, self.labels, offset=self.mean_t, want_mindist=want_mindist)
        acc = self.comm.sum_(acc).cpu().numpy()
        d = self.d
        return acc[: k * d].reshape(k, d), acc[k * d: k * d + k], float(acc[k * d + k]), int(round(acc[k * d + k + 1])), md


def _kmeans_plusplus(pts: _DevicePoints, k: int, rs: np.random.RandomState) -> np.ndarray:
    """sklearn _kmeans_plusplus (unit weights) on the centred data.  The distances to a new centre, the running minimum
    and the candidates' potentials are HIP passes over the resident shard (dcv_kmeanspp_update / _potentials); the host
    keeps what makes the draws reproducible: the RandomState stream and the sequential float64 cumsum + searchsorted
    over the closest-distance vector (8 bytes per point leave the device per centre, not the points).  Over several
    ranks that vector is gathered in rank (= frame) order, every rank runs the same seeded arithmetic and the candidate
    rows are fetched from their owners: identical centres everywhere and identical to the single-process run."""
    comm, dev, d = pts.comm, pts.dev, pts.d
    n = pts.n
    sizes = [int(v.item()) for v in comm.all_gather(torch.tensor([pts.n_local], dtype=torch.int64, device=dev))]
    start = int(sum(sizes[: comm.rank]))

    def rows(global_idx: np.ndarray) -> torch.Tensor:
        """Centred coordinates of the points with the given global indices (every rank gets all of them)."""
        gi = torch.as_tensor(np.asarray(global_idx, dtype=np.int64), device=dev)
        mine = (gi >= start) & (gi < start + pts.n_local)
        out = torch.zeros(len(gi), d, dtype=torch.float64, device=dev)
        if bool(mine.any()):
            out[mine] = pts.P[gi[mine] - start] - pts.mean_t
        return comm.sum_(out)

    closest = torch.empty(pts.n_local, dtype=torch.float64, device=dev)
    host = torch.empty(n, dtype=torch.float64, pin_memory=True)   # page-locked landing buffer of the distance vector
    centers = np.empty((k, d))
    trials = 2 + int(np.log(k))
    # first centre: random_state.choice(n, p=uniform) -- one uniform draw against the cumulative weights
    p_uniform = np.full(n, 1.0 / n)
    cid = rs.choice(n, p=p_uniform)
    c0 = rows([cid])
    centers[0] = c0[0].cpu().numpy()
    pot = float(comm.sum_(hip.kmeanspp_update(pts.P, c0[0].contiguous(), closest, True, offset=pts.mean_t)).item())
    for c in range(1, k):
        rand_vals = rs.uniform(size=trials) * pot
        host.copy_(comm.all_gather_rows(closest) if comm.active else closest)
        cand = np.searchsorted(np.cumsum(host.numpy(), dtype=np.float64), rand_vals)   # sklearn stable_cumsum: sequential float64
        np.clip(cand, None, n - 1, out=cand)
        cand_rows = rows(cand)
        pots = comm.sum_(hip.kmeanspp_potentials(pts.P, cand_rows, closest, offset=pts.mean_t)).cpu().numpy()
        best = int(np.argmin(pots))
        centers[c] = cand_rows[best].cpu().numpy()
        pot = float(comm.sum_(hip.kmeanspp_update(pts.P, cand_rows[best].contiguous(), closest, False, offset=pts.mean_t)).item())
    return centers


def _lloyd(pts: _DevicePoints, centers_init: np.ndarray):
    """sklearn _kmeans_single_lloyd: E+M passes until the labels stop changing or the squared
    centre shift drops below tol; a final E-step when stopped by tolerance; empty clusters are
    re-seeded with the points farthest from their centres."""
    k = centers_init.shape[0]
    centers = centers_init.copy()
    pts.labels.fill_(-1)
    strict = False
    n_iter = 0
    for it in range(KMEANS_MAX_ITER):
        n_iter = it + 1
        sums, counts, _, changed, md = pts.step(centers, want_mindist=False)
        empty = np.where(counts == 0)[0]
        if len(empty):
            # sklearn _relocate_empty_clusters_dense: the points farthest from their (old) centres become the
            # new centres of the empty clusters.  Candidates are merged over the ranks (values, owner, payload =
            # [old label | centred coordinates]); every rank then applies the same update to the global sums.
            _, _, _, _, md = pts.step(centers, want_mindist=True)   # distances to the assigned (old) centres
            topv, topi = torch.topk(md, min(len(empty), pts.n_local))   # this shard's candidates, farthest first
            payload = torch.cat([pts.labels[topi].to(torch.float64)[:, None], pts.P[topi] - pts.mean_t], dim=1)
            _, _, far = global_topk(topv, payload, len(empty), pts.comm)
            far = far.cpu().numpy()
            for e, row in zip(empty, far):
                old, x = int(row[0]), row[1:]
                sums[old] -= x
                sums[e] = x
                counts[e] = 1
                counts[old] -= 1
        new = sums / counts[:, None]
        shift_tot = float(((new - centers) ** 2).sum())
        centers = new
        if changed == 0:
            strict = True
            break
        if shift_tot <= pts.tol_abs:
            break
    if not strict:
        pts.step(centers)
    _, _, inertia, _, _ = pts.step(centers)   # inertia (and labels) against the final centres
    return pts.labels.clone(), inertia, centers, n_iter


def _same_clustering(a: torch.Tensor, b: torch.Tensor, k: int) -> bool:
    pairs = torch.unique(a.to(torch.int64) * k + b.to(torch.int64))
    return int(pairs.numel()) == int(torch.unique(a).numel()) == int(torch.unique(b).numel())


def kmeans_clustering(feature_matrix: np.ndarray, num_clusters: int, n_init: int,
                      initial_centroids: Optional[np.ndarray] = None, comm: Optional[Comm] = None, pts: Optional[_DevicePoints] = None):
    """KMeans(n_clusters, random_state=0, init='k-means++' | array, n_init).fit_predict on the
    GPU.  Returns (labels int32 NumPy of this rank's points, centres k x d float64)."""
    comm = comm or Comm()
    if pts is None:   # (optimize_clustering hands in the resident points: one upload for the whole scan over k)
        pts = _DevicePoints(feature_matrix, comm)
    rs = np.random.RandomState(0)
    if initial_centroids is not None:
        init = np.asarray(initial_centroids, dtype=np.float64) - pts.mean
        num_clusters, n_init = init.shape[0], 1
    else:
        init = None
    if not (1 <= num_clusters <= 64) or pts.d > 16:
        raise NotImplementedError(f"the HIP k-means kernel supports k <= 64, d <= 16 (got k={num_clusters}, d={pts.d})")
    logger.debug(f"Number of clusters: {num_clusters}")
    best = None
    for _ in range(n_init):
        c0 = init if init is not None else _kmeans_plusplus(pts, num_clusters, rs)
        labels, inertia, centers, n_iter = _lloyd(pts, c0)
        if best is None or (inertia < best[1] and not _same_clustering(labels, best[0], num_clusters)):
            best = (labels, inertia, centers, n_iter)
    labels, inertia, centers, _ = best
    return labels.cpu().numpy(), centers + pts.mean


def cluster_data(features: np.ndarray, settings: Dict, initial_centroids: np.ndarray = None, pts: Optional[_DevicePoints] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Cluster with the algorithm named in `settings` (defaults filled in place, as the reference
    does, statistics.py:134-142)."""
    settings["algorithm"] = settings.get("algorithm", "kmeans")
    settings["num_clusters"] = settings.get("num_clusters", 10)
    settings["n_init"] = settings.get("n_init", 10)
    settings["min_cluster_size"] = settings.get("min_cluster_size", int(0.1 * features.shape[0]))
    settings["min_samples"] = settings.get("min_samples", max(int(0.001 * features.shape[0]), 1))
    settings["cluster_selection_epsilon"] = settings.get("cluster_selection_epsilon", 0)
    settings["linkage"] = settings.get("linkage", "complete")
    settings["max_cluster_size"] = settings.get("max_cluster_size")
    settings["cluster_selection_method"] = settings.get("cluster_selection_method", "eom")
    algo = settings["algorithm"]
    if algo == "kmeans":
        return kmeans_clustering(features, settings["num_clusters"], settings["n_init"], initial_centroids, pts=pts)
    if algo == "hierarchical":
        from sklearn.cluster import AgglomerativeClustering

        labels = AgglomerativeClustering(n_clusters=settings["num_clusters"], distance_threshold=None,
                                         linkage=settings["linkage"]).fit_predict(features)
        cents = np.stack([features[labels == i].mean(axis=0) for i in range(len(np.unique(labels)))])
        return labels, cents
    if algo == "hdbscan":
        from sklearn.cluster import HDBSCAN

        hdb = HDBSCAN(min_cluster_size=settings["min_cluster_size"], min_samples=settings["min_samples"], store_centers="centroid",
                      cluster_selection_epsilon=settings["cluster_selection_epsilon"], max_cluster_size=settings["max_cluster_size"],
                      cluster_selection_method=settings["cluster_selection_method"], allow_single_cluster=False)
        hdb.fit(features)
        return hdb.labels_, hdb.centroids_
    raise Exception(f"clustering algorithm {algo} not implemented")


def clustering_scores(features: np.ndarray, labels: np.ndarray, comm: Optional[Comm] = None,
                      silhouette_max_points: Optional[int] = None, P_dev: Optional[torch.Tensor] = None) -> Tuple[float, float, float]:
    """(Calinski-Harabasz, Davies-Bouldin, silhouette) of a labelling on the GPU, the definitions of
    sklearn.metrics the reference calls (statistics.py:73-75).  `features` / `labels` are this rank's
    block of frames.  CH and DB are two streaming passes (label sums -> means, then dispersions);
    the silhouette is the exact all-pairs form, O(n^2 d) float64: `silhouette_max_points` (opt-in, a
    behaviour change with respect to the reference) evaluates it on an evenly strided subset of the
    query points against all points."""
    comm = comm or Comm()
    dev = _device()
    P = P_dev if P_dev is not None else torch.from_numpy(np.ascontiguousarray(features, dtype=np.float64)).to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
    d = P.shape[1]
    k = int(comm.max_(lab.max().reshape(1).to(torch.int64)).item()) + 1 if comm.active else int(lab.max().item()) + 1
    if not (1 <= k <= 64) or d > 16:
        raise NotImplementedError(f"the HIP score kernels support k <= 64, d <= 16 (got k={k}, d={d})")
    acc = comm.sum_(hip.label_stats(P, lab, k)).cpu().numpy()
    counts = acc[k * d: k * d + k]
    n = counts.sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        means = acc[: k * d].reshape(k, d) / counts[:, None]
    means = np.nan_to_num(means)
    acc = comm.sum_(hip.label_stats(P, lab, k, centers=torch.from_numpy(means).to(dev))).cpu().numpy()
    ss = acc[k * d + k: k * d + 2 * k]
    sd = acc[k * d + 2 * k: k * d + 3 * k]
    present = counts > 0
    n_labels = int(present.sum())
    # Calinski-Harabasz (sklearn: extra_disp * (n - k) / (intra_disp * (k - 1)), 1.0 when intra_disp == 0)
    mean_all = (means * counts[:, None]).sum(axis=0) / n
    extra = float((counts * ((means - mean_all) ** 2).sum(axis=1)).sum())
    intra = float(ss.sum())
    ch = 1.0 if intra == 0.0 else extra * (n - n_labels) / (intra * (n_labels - 1.0))
    # Davies-Bouldin (sklearn: mean over clusters of max_j (s_i + s_j) / d_ij, 0/0 -> 0)
    mp = means[present]
    sp = sd[present] / counts[present]
    cd = np.sqrt(((mp[:, None, :] - mp[None, :, :]) ** 2).sum(axis=2))
    if np.allclose(sp, 0) or np.allclose(cd, 0):
        db = 0.0
    else:
        cd[cd == 0] = np.inf
        db = float(np.mean(np.max((sp[:, None] + sp[None, :]) / cd, axis=1)))
    # silhouette: all points sorted by cluster (gathered over the ranks), this rank's points as queries
    P_all = comm.all_gather_rows(P)
    lab_all = comm.all_gather_rows(lab)
    order = torch.argsort(lab_all, stable=True)
    P_sorted = P_all[order].contiguous()
    start = torch.zeros(k + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(torch.bincount(lab_all.clamp(min=0).to(torch.int64), minlength=k)[:k], dim=0)
    Q, ql = P, lab
    if silhouette_max_points is not None and n > silhouette_max_points:
        stride = int(np.ceil(n / silhouette_max_points))
        Q, ql = P[::stride].contiguous(), lab[::stride].contiguous()
    tot = torch.zeros(1, dtype=torch.float64, device=dev)
    nq = 0
    chunk = max(1, (1 << 26) // max(k, 1))           # bound the nq x k float64 scratch to 512 MiB
    for b in range(0, Q.shape[0], chunk):
        S = hip.cluster_dist_sums(Q[b:b + chunk], P_sorted, start)
        tot += hip.silhouette_sum(S, ql[b:b + chunk], start)
        nq += min(chunk, Q.shape[0] - b)
    tot = comm.sum_(tot)
    nq = comm.sum_scalar(float(nq), device=dev)
    return float(ch), float(db), float(tot.item() / nq)


def optimize_clustering(features: np.ndarray, settings: Dict):
    """k in search_interval (inclusive): cluster, Calinski-Harabasz / Davies-Bouldin / silhouette,
    min-max normalise each list, best (CH - DB + Sil) / 3 (reference :17-110).  The scores stay
    on scikit-learn (silhouette is O(N^2): SURVEY.md section 8 f4)."""
    if settings["algorithm"] in ("kmeans", "hierarchical"):
        from sklearn.metrics import calinski_harabasz_score, davies_bouldin_score, silhouette_score

        lo, hi = settings.get("search_interval", [2, 15])
        ks = list(range(lo, hi + 1))
        ch, db, si, results = [], [], [], []
        # k-means: the points go to the GPU once for the whole scan (each k used to upload them twice: clustering, then scores)
        pts = _DevicePoints(features, Comm()) if settings["algorithm"] == "kmeans" else None
        for N in ks:
            settings["num_clusters"] = N
            labels, centroids = cluster_data(features, settings, pts=pts)
            if settings["algorithm"] == "kmeans":   # scores on the GPU (same definitions: clustering_scores)
                c_, d_, s_ = clustering_scores(features, labels, silhouette_max_points=settings.get("silhouette_max_points"), P_dev=pts.P)
            else:
                c_, d_, s_ = (calinski_harabasz_score(features, labels), davies_bouldin_score(features, labels),
                              silhouette_score(features, labels))
            ch.append(c_)
            db.append(d_)
            si.append(s_)
            results.append((labels, centroids))
        with np.errstate(invalid="ignore", divide="ignore"):
            ch = (ch - np.min(ch)) / (np.max(ch) - np.min(ch))
            db = (db - np.min(db)) / (np.max(db) - np.min(db))
            si = (si - np.min(si)) / (np.max(si) - np.min(si))
        score = (np.array(ch) - np.array(db) + np.array(si)) / 3
        best = int(np.argmax(score))
        logger.info(f"Best number of clusters: {ks[best]}")
        cluster_labels, centroids = results[best]
    elif settings["algorithm"] == "hdbscan":
        cluster_labels, centroids = cluster_data(features, settings)
    else:
        raise Exception(f"clustering algorithm {settings['algorithm']} not implemented")
    if len(centroids) == 0:
        logger.warning("No clusters found using the provided settings. Try different settings or a different algorithm")
    return cluster_labels, centroids


def find_centroids(data: pd.DataFrame, centroids: np.ndarray, clustering_features: list, comm: Optional[Comm] = None,
                   row_offset: int = 0) -> pd.DataFrame:
    """Adds a boolean 'centroid' column marking, for every centroid, the globally nearest sample
    (np.linalg.norm, first index on ties) -- one HIP pass per centroid (reference :337-379)."""
    if len(centroids) == 0:
        logger.warning("No centroids found")
        return pd.DataFrame()
    if len(centroids[0]) != len(clustering_features):
        logger.error("  The dimension of the centroids is not the same as the dimension of the used features for clustering.\n")
        sys.exit(1)
    comm = comm or Comm()
    dev = _device()
    P = torch.from_numpy(np.ascontiguousarray(data.loc[:, clustering_features].values, dtype=np.float64)).to(dev)
    C = torch.from_numpy(np.ascontiguousarray(centroids, dtype=np.float64)).to(dev)
    dist, rows = hip.nearest_rows(P, C, row_offset=row_offset)
    _, rows = reduce_nearest(dist, rows, comm)
    data["centroid"] = False
    for r in rows.cpu().numpy():
        local = int(r) - row_offset
        if 0 <= local < len(data):
            data.at[data.index[local], "centroid"] = True
    return data


def assign_closest_cluster(train_points: np.ndarray, train_labels: np.ndarray, sup_points: np.ndarray) -> np.ndarray:
    """Label of the nearest training point for each supplementary point
    (TrajClusterWorkflow.assign_closest_cluster, traj_cluster_workflow.py:207-238)."""
    dev = _device()
    nn = hip.nearest_point(torch.from_numpy(np.ascontiguousarray(train_points, dtype=np.float64)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(sup_points, dtype=np.float64)).to(dev))
    return np.asarray(train_labels)[nn.cpu().numpy()]


# ------------------------------------------------------------------------------------------- free-energy surface
KB_KJ_MOL = 0.0083144621   # kJ / (mol K), mlcolvar's default units


def compute_fes(data: np.ndarray, temperature: float = 300.0, bandwidth: float = 0.05, num_bins: int = 150, blocks: int = 1,
                eps: float = 1e-10, bounds=None, comm: Optional[Comm] = None):
    """Free-energy surface of a 1-D or 2-D projected trajectory as the reference's figures.plot_fes obtains it from
    mlcolvar.utils.fes.compute_fes(data, temp, backend="KDEpy", num_samples=num_bins, bandwidth, blocks, eps, bounds)
    (figures.py:95-98): binned Gaussian kernel density on a num_bins^d grid, F = -kB T log(density + eps), shifted to a
    zero minimum; with `blocks` > 1 the mean over consecutive blocks of frames and its standard error.
    The pass over the frames (linear binning) is a HIP kernel; the convolution of the small grid with the Gaussian and the
    logarithm run on the host.  Returns (fes, grid, bounds, error) with grid = the per-dimension node coordinates.
    [mlcolvar / KDEpy are not under /root/reference nor installed: restated from their published algorithm, parity
    unpinned by any reference fixture; checked against the same algorithm in NumPy and against the exact Gaussian KDE.]"""
    comm = comm or Comm()
    X = np.asarray(data, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    n, d = X.shape
    if d not in (1, 2):
        raise ValueError("compute_fes handles 1-D and 2-D data")
    if bounds is None:   # mlcolvar: data range widened by a small offset
        bounds = [(float(X[:, c].min()) - 1e-3, float(X[:, c].max()) + 1e-3) for c in range(d)]
    lo, hi = [b[0] for b in bounds], [b[1] for b in bounds]
    grid = [np.linspace(lo[c], hi[c], num_bins) for c in range(d)]
    kbt = KB_KJ_MOL * float(temperature)
    dev = _device()
    P = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
    # Gaussian kernel sampled at the node spacing, per dimension (separable), normalised to unit sum
    kernels = []
    for c in range(d):
        h = (hi[c] - lo[c]) / (num_bins - 1)
        half = int(np.ceil(6.0 * bandwidth / h))
        t = np.arange(-half, half + 1) * h
        kernels.append(np.exp(-0.5 * (t / bandwidth) ** 2) / (bandwidth * np.sqrt(2.0 * np.pi)))
    blocks = max(1, int(blocks))
    size = n // blocks
    fes_blocks = []
    for b in range(blocks):
        part = P[b * size: (b + 1) * size if b < blocks - 1 else n]
        w, _ = hip.linear_binning(part, list(range(d)), lo, hi, num_bins)
        count = torch.tensor([float(part.shape[0])], dtype=torch.float64, device=dev)
        if comm.active:
            comm.sum_(w)
            comm.sum_(count)
        dens = w.cpu().numpy() / float(count.item())
        for c in range(d):   # density(node) = sum_nodes weight * K(node - node'): convolution along each axis
            dens = np.apply_along_axis(lambda v: np.convolve(v, kernels[c], mode="same"), c, dens)
        fes_blocks.append(-kbt * np.log(dens + eps))
    fes_blocks = np.stack(fes_blocks)
    fes = fes_blocks.mean(axis=0)
    error = fes_blocks.std(axis=0, ddof=1) / np.sqrt(blocks) if blocks > 1 else np.zeros_like(fes)
    fes = fes - fes.min()
    if d == 2:
        fes, error = fes.T, error.T   # mlcolvar evaluates on np.meshgrid(x, y): rows = second coordinate
    return fes, (grid[0] if d == 1 else np.stack(np.meshgrid(*grid))), np.array(bounds), error
