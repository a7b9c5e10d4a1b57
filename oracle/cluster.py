"""Oracle: k-means clustering, k selection and centroid search.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Two layers:

* ``kmeans_reference_backend`` calls scikit-learn exactly as the reference does
  (modules/statistics/statistics.py:183-197) -- scikit-learn IS the reference's backend
  for this row, so this is the pinning oracle when the reference module itself cannot
  travel (it never leaves the build container);
* ``kmeans_restated`` is a NumPy restatement of the algorithm scikit-learn executes
  (KMeans.fit -> k-means++ -> Lloyd), the form the HIP path mirrors.
"""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------- backend call
def kmeans_reference_backend(X: np.ndarray, num_clusters: int, n_init: int, initial_centroids=None):
    """statistics.kmeans_clustering (statistics.py:159-197)."""
    from sklearn.cluster import KMeans

    init = "k-means++" if initial_centroids is None else initial_centroids
    if initial_centroids is not None:
        num_clusters = initial_centroids.shape[0]
    km = KMeans(n_clusters=num_clusters, random_state=0, init=init, n_init=n_init)
    labels = km.fit_predict(X)
    return labels, km.cluster_centers_, km.inertia_, km.n_iter_


# ----------------------------------------------------------------------------- restatement
def _sq_dists(C: np.ndarray, X: np.ndarray, xsq: np.ndarray) -> np.ndarray:
    """sklearn.metrics.pairwise._euclidean_distances(C, X, Y_norm_squared=xsq, squared=True)."""
    d = -2.0 * (C @ X.T)
    d += (C * C).sum(axis=1)[:, None]
    d += xsq[None, :]
    np.maximum(d, 0, out=d)
    return d


def kmeans_plusplus(X: np.ndarray, k: int, xsq: np.ndarray, rs: np.random.RandomState):
    """sklearn.cluster._kmeans._kmeans_plusplus with unit sample weights (Appendix A.8 step 4)."""
    n = X.shape[0]
    centers = np.empty((k, X.shape[1]), dtype=X.dtype)
    n_local_trials = 2 + int(np.log(k))
    w = np.ones(n, dtype=X.dtype)
    cid = rs.choice(n, p=w / w.sum())
    indices = np.full(k, -1, dtype=int)
    centers[0] = X[cid]
    indices[0] = cid
    closest = _sq_dists(centers[0, None], X, xsq)
    pot = closest @ w
    for c in range(1, k):
        rand_vals = rs.uniform(size=n_local_trials) * pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64).ravel(), rand_vals)
        np.clip(cand, None, closest.size - 1, out=cand)
        dc = _sq_dists(X[cand], X, xsq)
        np.minimum(closest, dc, out=dc)
        cpot = dc @ w.reshape(-1, 1)
        best = np.argmin(cpot)
        pot = cpot[best]
        closest = dc[best]
        centers[c] = X[cand[best]]
        indices[c] = cand[best]
    return centers, indices


def lloyd(X: np.ndarray, centers_init: np.ndarray, tol_abs: float, max_iter: int = 300):
    """sklearn _kmeans_single_lloyd on dense float64 data, unit weights (Appendix A.8 step 5):
    argmin_j(||c_j||^2 - 2 x.c_j) with first-minimum tie-break, empty clusters relocated to
    the farthest points, strict-convergence / tolerance stop, final E-step re-label."""
    k = centers_init.shape[0]
    centers = centers_init.copy()
    labels_old = np.full(X.shape[0], -1, dtype=np.int32)
    strict = False
    n_iter = 0
    for it in range(max_iter):
        n_iter = it + 1
        pw = (centers * centers).sum(axis=1)[None, :] - 2.0 * (X @ centers.T)
        labels = pw.argmin(axis=1).astype(np.int32)
        sums = np.zeros_like(centers)
        np.add.at(sums, labels, X)
        counts = np.bincount(labels, minlength=k).astype(X.dtype)
        empty = np.where(counts == 0)[0]
        if len(empty):
            dist = ((X - centers[labels]) ** 2).sum(axis=1)
            far = np.argpartition(dist, -len(empty))[: -len(empty) - 1: -1]
            for e, f in zip(empty, far):
                old = labels[f]
                sums[old] -= X[f]
                sums[e] = X[f]
                counts[e] = 1
                counts[old] -= 1
        new = sums / counts[:, None]
        shift = np.sqrt(((new - centers) ** 2).sum(axis=1))
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol_abs:
            break
        labels_old = labels
    if not strict:
        pw = (centers * centers).sum(axis=1)[None, :] - 2.0 * (X @ centers.T)
        labels = pw.argmin(axis=1).astype(np.int32)
    inertia = float(((X - centers[labels]) ** 2).sum())
    return labels, inertia, centers, n_iter


def _same_clustering(a, b, k):
    mapping = np.full(k, -1, dtype=np.int64)
    for x, y in zip(a, b):
        if mapping[x] == -1:
            mapping[x] = y
        elif mapping[x] != y:
            return False
    return True


def kmeans_restated(X: np.ndarray, num_clusters: int, n_init: int, initial_centroids=None,
                    tol: float = 1e-4, max_iter: int = 300):
    """KMeans(n_clusters, random_state=0, init, n_init).fit (Appendix A.8)."""
    X = np.array(X, dtype=np.float64, copy=True)
    rs = np.random.RandomState(0)
    init = None
    if initial_centroids is not None:
        init = np.array(initial_centroids, dtype=np.float64, copy=True)
        num_clusters = init.shape[0]
        n_init = 1
    mean = X.mean(axis=0)
    X -= mean
    if init is not None:
        init -= mean
    xsq = (X * X).sum(axis=1)
    tol_abs = np.mean(np.var(X, axis=0)) * tol
    best = None
    for _ in range(n_init):
        c0 = init if init is not None else kmeans_plusplus(X, num_clusters, xsq, rs)[0]
        labels, inertia, centers, n_iter = lloyd(X, c0, tol_abs, max_iter)
        if best is None or (inertia < best[1] and not _same_clustering(labels, best[0], num_clusters)):
            best = (labels, inertia, centers, n_iter)
    labels, inertia, centers, n_iter = best
    return labels, centers + mean, inertia, n_iter


# ----------------------------------------------------------------------------- centroids
def find_centroid_rows(P: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """statistics.find_centroids (statistics.py:337-379): for every centroid the row index of
    the globally nearest sample (np.argmin of the Euclidean norm => first index on ties)."""
    rows = []
    for c in centroids:
        d = np.linalg.norm(P - c, axis=1)
        rows.append(int(np.argmin(d)))
    return np.asarray(rows, dtype=np.int64)


def nearest_neighbour_labels(train: np.ndarray, labels: np.ndarray, sup: np.ndarray) -> np.ndarray:
    """TrajClusterWorkflow.assign_closest_cluster (traj_cluster_workflow.py:207-238):
    label of the 1-nearest training point for each supplementary point."""
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(n_neighbors=1).fit(train)
    _, idx = nn.kneighbors(sup)
    return labels[idx[:, 0]]


# ----------------------------------------------------------------------------- k selection
def combined_scores(P: np.ndarray, label_sets) -> np.ndarray:
    """The score of optimize_clustering (statistics.py:73-93): min-max normalised
    (CH - DB + silhouette) / 3 over the candidate clusterings."""
    from sklearn.metrics import calinski_harabasz_score, davies_bouldin_score, silhouette_score

    ch = np.array([calinski_harabasz_score(P, l) for l in label_sets])
    db = np.array([davies_bouldin_score(P, l) for l in label_sets])
    si = np.array([silhouette_score(P, l) for l in label_sets])
    with np.errstate(invalid="ignore", divide="ignore"):
        ch = (ch - ch.min()) / (ch.max() - ch.min())
        db = (db - db.min()) / (db.max() - db.min())
        si = (si - si.min()) / (si.max() - si.min())
    return (ch - db + si) / 3
