"""Parity of the HIP kernels (through the C-ABI) against the CPU oracle / float64 NumPy.
Run on the GPU box: python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

from oracle import cluster as oc
from oracle import linear as ol

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(a).cuda()


def rand_matrix(n, F, seed, scramble=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.standard_normal((n, F)).astype(np.float32)
    if scramble:
        X = X * rng.uniform(0.1, 10, F).astype(np.float32) + rng.uniform(-5, 5, F).astype(np.float32)
    return X


@pytest.mark.parametrize("n,F", [(164, 54), (10000, 128), (5000, 512), (3001, 1024), (700, 2048), (1, 8), (5, 3)])
def test_col_stats(n, F):
    from deep_cartograph_amd import hip

    X = rand_matrix(n, F, 1)
    raw = hip.col_stats_raw(dev(X)).cpu().numpy()
    X64 = X.astype(np.float64)
    np.testing.assert_allclose(raw[0], X64.sum(0), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(raw[1], (X64 * X64).sum(0), rtol=1e-12)
    np.testing.assert_array_equal(raw[2], X64.min(0))
    np.testing.assert_array_equal(raw[3], X64.max(0))
    if n > 1:
        st = hip.finalize_stats(torch.from_numpy(raw), n)
        ref = ol.feature_stats(X)
        # tolerance: pandas accumulates in float32, the kernel in float64
        np.testing.assert_allclose(st["mean"], ref["mean"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(st["std"], ref["std"], rtol=2e-5)
        np.testing.assert_array_equal(st["min"], ref["min"])
        np.testing.assert_array_equal(st["max"], ref["max"])


@pytest.mark.parametrize("n,F", [(164, 54), (4096, 256), (1000, 100)])
def test_normalize_bit_exact(n, F):
    from deep_cartograph_amd import hip

    X = rand_matrix(n, F, 2)
    st = ol.feature_stats(X)
    m, r = ol.prepare_normalization(st, "mean_std")
    ref = ol.normalize(X, m, r)
    Xd = dev(X)
    out = hip.normalize(Xd, dev(m.astype(np.float32)), dev(r.astype(np.float32)))
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    hip.normalize(Xd, dev(m.astype(np.float32)), dev(r.astype(np.float32)), out=Xd)  # in place, as the reference
    np.testing.assert_array_equal(Xd.cpu().numpy(), ref)


GEMM_SHAPES = [(128, 128, 64), (300, 200, 100), (64, 4, 128), (4, 128, 1000), (257, 33, 31), (1000, 256, 512),
               (129, 130, 33), (32, 32, 2), (512, 2, 7), (2, 512, 40)]


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_exact_on_integers(mode, M, N, K):
    """Small-integer operands make every fp32 product and partial sum exact, so the MFMA
    engine must agree bit for bit with an int64 product whatever the tile / k-slot order;
    B is asymmetric, so a transposed C write cannot pass."""
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(M * 7 + N * 3 + K))
    A = rng.integers(-4, 5, size=(M, K))
    B = rng.integers(-4, 5, size=(K, N))
    ref = (A @ B).astype(np.float32)
    Af = A.astype(np.float32)
    Bf = B.astype(np.float32)
    if mode == "nt":
        out = hip.gemm("nt", dev(Af), dev(np.ascontiguousarray(Bf.T)))
    elif mode == "nn":
        out = hip.gemm("nn", dev(Af), dev(Bf))
    else:
        out = hip.gemm("tn", dev(np.ascontiguousarray(Af.T)), dev(Bf))
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_random_fp32(mode):
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(5))
    M, N, K = 384, 256, 640
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    if mode == "nt":
        out = hip.gemm("nt", dev(A), dev(np.ascontiguousarray(B.T)))
    elif mode == "nn":
        out = hip.gemm("nn", dev(A), dev(B))
    else:
        out = hip.gemm("tn", dev(np.ascontiguousarray(A.T)), dev(B))
    # fp32 fma chain: error <= ~1.5e-7 * sum|a||b| (guide: FP32-input MFMA numerics)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    assert np.all(np.abs(out.cpu().numpy() - ref) <= bound + 1e-6)


def test_gemm_modes_split_vs_native():
    """The two arithmetic flavours of the product engine (dcv_set_gemm_mode): FP32-input MFMA and FP32-accurate split
    products on the BF16 pipe.  Both meet the fp32 error bound against float64 on zero-mean data; on a long same-sign
    accumulation (K = 8192) the BF16 pipe's truncating accumulator shows its bias -- bounded here at 4e-5 relative,
    which is why the covariance kernels never use it -- while the native path stays unbiased."""
    from deep_cartograph_amd import hip

    start = hip.get_gemm_mode()
    try:
        rng = np.random.Generator(np.random.PCG64(8))
        A = rng.standard_normal((256, 512)).astype(np.float32)
        B = rng.standard_normal((192, 512)).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T)
        outs = {}
        for mode in ("native", "split"):
            hip.set_gemm_mode(mode)
            assert hip.get_gemm_mode() == mode
            outs[mode] = hip.gemm("nt", dev(A), dev(B)).cpu().numpy()
            assert np.all(np.abs(outs[mode] - ref) <= bound + 1e-6), mode
        assert not np.array_equal(outs["native"], outs["split"])       # the switch does switch
        Ap = (rng.random((8192, 128)) + 0.1).astype(np.float32)        # K = 8192 rows of positive numbers (TN form)
        Bp = (rng.random((8192, 128)) + 0.1).astype(np.float32)
        refp = Ap.astype(np.float64).T @ Bp.astype(np.float64)
        rel = {}
        for mode in ("native", "split"):
            hip.set_gemm_mode(mode)
            got = hip.gemm("tn", dev(Ap), dev(Bp)).cpu().numpy()
            rel[mode] = (got - refp) / refp
        assert np.max(np.abs(rel["native"])) < 2e-5 and abs(rel["native"].mean()) < 2e-6
        assert np.max(np.abs(rel["split"])) < 4e-5
    finally:
        hip.set_gemm_mode(start)


@pytest.mark.parametrize("n,F,lag,shift", [(5000, 64, 3, False), (3000, 54, 1, True), (40000, 256, 10, False),
                                           (20000, 128, 0, True), (1500, 300, 7, True), (40, 8, 39, False)])
def test_lagged_cov_vs_float64(n, F, lag, shift):
    from deep_cartograph_amd import hip

    X = rand_matrix(n, F, 3, scramble=shift)
    P = n - lag
    sh = X.mean(0).astype(np.float32) if shift else None
    raw = hip.lagged_cov_raw(dev(X), P, lag, dev(sh) if shift else None).cpu().numpy()
    Z = X.astype(np.float32) - (sh if shift else 0)
    Z = Z.astype(np.float64)  # the kernel subtracts in fp32
    zt, zl = Z[:P], Z[lag:lag + P]
    a, b = zt.sum(0), zl.sum(0)
    A = zt.T @ zt
    B = zt.T @ zl
    scale = np.sqrt(np.outer(np.diag(A), np.diag(A)))
    np.testing.assert_allclose(raw[:F], a, rtol=1e-9, atol=1e-6 * P)
    if lag:
        np.testing.assert_allclose(raw[F:2 * F], b, rtol=1e-9, atol=1e-6 * P)
    got_A = raw[2 * F:2 * F + F * F].reshape(F, F)
    got_B = raw[2 * F + F * F:].reshape(F, F)
    # fp32 products and chunked fp32 accumulation: relative to the diagonal scale
    assert np.max(np.abs(got_A - A) / scale) < 3e-6
    if lag:
        assert np.max(np.abs(got_B - B) / scale) < 3e-6
    else:
        assert np.all(got_B == 0)
    # deterministic
    raw2 = hip.lagged_cov_raw(dev(X), P, lag, dev(sh) if shift else None).cpu().numpy()
    np.testing.assert_array_equal(raw, raw2)


@pytest.mark.parametrize("n,F,d", [(164, 54, 2), (10000, 256, 4), (3000, 512, 3), (777, 100, 8), (5000, 1024, 16), (100, 64, 1)])
def test_project_linear(n, F, d):
    from deep_cartograph_amd import hip

    X = rand_matrix(n, F, 4)
    st = ol.feature_stats(X)
    m, r = ol.prepare_normalization(st, "mean_std")
    rng = np.random.Generator(np.random.PCG64(9))
    W = (rng.standard_normal((F, d)) / np.sqrt(F)).astype(np.float32)
    Xn = ol.normalize(X, m, r)
    cm, cr = ol.linear_cv_norm(Xn, W)
    ref = ol.project_linear(X, W, cm, cr, m, r)
    out, mm = hip.project_linear(dev(X), dev(W), fmean=dev(m.astype(np.float32)), frange=dev(r.astype(np.float32)),
                                 cvmean=dev(cm.astype(np.float32)), cvrange=dev(cr.astype(np.float32)), want_minmax=True)
    out = out.cpu().numpy()
    np.testing.assert_allclose(out, ref, atol=2e-5)  # |out| <= 1 after min-max normalisation
    np.testing.assert_allclose(mm.cpu().numpy(), np.stack([out.min(0), out.max(0)]), atol=0)
    # raw projection of already-normalised data + extrema only (normalize_cv path)
    _, mm2 = hip.project_linear(dev(Xn), dev(W), want_out=False, want_minmax=True)
    Praw = Xn.astype(np.float64) @ W.astype(np.float64)
    np.testing.assert_allclose(mm2.cpu().numpy(), np.stack([Praw.min(0), Praw.max(0)]), rtol=1e-5, atol=1e-5)


def test_kmeans_step_and_nearest(golden_cluster):
    from deep_cartograph_amd import hip

    for tag in ("syn_a", "syn_b", "syn_c"):
        P = golden_cluster[f"{tag}.points"]
        C = golden_cluster[f"{tag}.init"]
        mean = P.mean(0)
        Pd = dev(P)
        labels = torch.full((P.shape[0],), -1, dtype=torch.int32, device="cuda")
        acc, md = hip.kmeans_step(Pd, dev(C - mean), labels, offset=dev(mean), want_mindist=True)
        acc = acc.cpu().numpy()
        Xc = P - mean
        Cc = C - mean
        pw = (Cc * Cc).sum(1)[None, :] - 2.0 * (Xc @ Cc.T)
        ref_lab = pw.argmin(1)
        np.testing.assert_array_equal(labels.cpu().numpy(), ref_lab)
        k, d = C.shape
        sums = np.zeros((k, d))
        np.add.at(sums, ref_lab, Xc)
        np.testing.assert_allclose(acc[:k * d].reshape(k, d), sums, rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(acc[k * d:k * d + k], np.bincount(ref_lab, minlength=k))
        ref_in = ((Xc - Cc[ref_lab]) ** 2).sum()
        np.testing.assert_allclose(acc[k * d + k], ref_in, rtol=1e-12)
        assert acc[k * d + k + 1] == P.shape[0]
        np.testing.assert_allclose(md.cpu().numpy(), ((Xc - Cc[ref_lab]) ** 2).sum(1), rtol=1e-12, atol=1e-15)
        # nearest sample of every centroid: bit-exact row indices vs np.linalg.norm / np.argmin
        cents = golden_cluster[f"{tag}.pp_centroids"]
        _, rows = hip.nearest_rows(Pd, dev(cents))
        np.testing.assert_array_equal(rows.cpu().numpy(), oc.find_centroid_rows(P, cents))
        # 1-NN label transfer
        sup = P[::37] + 1e-3
        nn = hip.nearest_point(Pd, dev(sup)).cpu().numpy()
        d2 = ((sup[:, None, :] - P[None, :, :]) ** 2).sum(-1)
        np.testing.assert_array_equal(nn, d2.argmin(1))


@pytest.mark.parametrize("d,k,n", [(4, 6, 200_003), (2, 3, 70_001), (4, 8, 63), (2, 1, 5_000), (4, 9, 30_000)])
def test_kmeans_pass_through_the_lds_ring(d, k, n):
    """The plain Lloyd pass (no per-point distances asked for) streams d = 2 / 4 points through the per-wave LDS-DMA ring
    (kmeans.hip: ring_stream): labels bit-exact against the sklearn formula in NumPy (first minimum wins), sums / counts /
    inertia / changed against NumPy, the same answers as the register pass (which reports the distances), ragged tails (n not
    a multiple of 64, fewer points than one chunk), an offset, k = 9 (the register kernel: more than 8 clusters)."""
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(1000 + 10 * d + k))
    cent = rng.uniform(-1, 1, (k, d))
    P = np.round(cent[rng.integers(0, k, n)] + 0.15 * rng.standard_normal((n, d)), 4)
    C = P[rng.choice(n, k, replace=False)] + 1e-3
    mean = P.mean(0)
    Pd = dev(P)
    lab_ring = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    lab_reg = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    acc_ring, none = hip.kmeans_step(Pd, dev(C - mean), lab_ring, offset=dev(mean))
    acc_reg, md = hip.kmeans_step(Pd, dev(C - mean), lab_reg, offset=dev(mean), want_mindist=True)
    assert none is None
    Xc, Cc = P - mean, C - mean
    ref_lab = ((Cc * Cc).sum(1)[None, :] - 2.0 * (Xc @ Cc.T)).argmin(1)
    np.testing.assert_array_equal(lab_ring.cpu().numpy(), ref_lab)
    np.testing.assert_array_equal(lab_reg.cpu().numpy(), ref_lab)
    sums = np.zeros((k, d))
    np.add.at(sums, ref_lab, Xc)
    for acc in (acc_ring.cpu().numpy(), acc_reg.cpu().numpy()):
        np.testing.assert_allclose(acc[:k * d].reshape(k, d), sums, rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(acc[k * d:k * d + k], np.bincount(ref_lab, minlength=k))
        np.testing.assert_allclose(acc[k * d + k], ((Xc - Cc[ref_lab]) ** 2).sum(), rtol=1e-12)
        assert acc[k * d + k + 1] == n
    again, _ = hip.kmeans_step(Pd, dev(C - mean), lab_ring, offset=dev(mean))      # idempotent, bit for bit
    again = again.cpu().numpy()
    assert again[k * d + k + 1] == 0
    np.testing.assert_array_equal(again[:k * d + k + 1], acc_ring.cpu().numpy()[:k * d + k + 1])


def test_nearest_rows_wide_d():
    """numpy's pairwise summation changes form at d >= 8."""
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(3))
    for d in (8, 11, 16):
        P = np.round(rng.uniform(-1, 1, (4000, d)), 4)
        C = np.round(rng.uniform(-1, 1, (5, d)), 3)
        _, rows = hip.nearest_rows(dev(P), dev(C))
        np.testing.assert_array_equal(rows.cpu().numpy(), oc.find_centroid_rows(P, C))


@pytest.mark.parametrize("d,k", [(1, 3), (2, 6), (3, 11), (4, 6), (4, 17)])
def test_nearest_rows_one_pass_narrow_d(d, k):
    """d <= 4: ONE pass over the points for every chunk of 8 centroids (nearest_rows_multi_kernel).  Rows bit-exact against
    np.linalg.norm / argmin incl. exact ties (duplicated points: the first index wins) and centroids that ARE points; the
    per-centroid kernel (DCV_NEAREST_PER_CENTROID=1 in a fresh process would select it) is covered by the wide-d test."""
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(100 + 10 * d + k))
    P = np.round(rng.uniform(-1, 1, (150_001, d)), 4)      # %.4f CSV values, as the pipeline hands them over
    P[70_000:70_050] = P[10_000:10_050]                   # exact duplicates: ties in every distance
    C = np.round(rng.uniform(-1, 1, (k, d)), 3)
    C[0] = P[70_010]                                       # distance 0, reached at two rows: row 10 010 must win
    dist, rows = hip.nearest_rows(dev(P), dev(C))
    ref = oc.find_centroid_rows(P, C)
    np.testing.assert_array_equal(rows.cpu().numpy(), ref)
    assert rows[0].item() == 10_010
    np.testing.assert_array_equal(dist.cpu().numpy(), np.linalg.norm(P[ref] - C, axis=1))
    # a row offset (frame-sharded callers) shifts the reported rows only
    _, rows_o = hip.nearest_rows(dev(P), dev(C), row_offset=1_000_000)
    np.testing.assert_array_equal(rows_o.cpu().numpy(), ref + 1_000_000)


@pytest.mark.parametrize("d,bins,blocks", [(1, 150, 1), (2, 100, 1), (2, 64, 4), (1, 5000, 1)])
def test_fes_binned_kde_matches_oracle(d, bins, blocks):
    """f4, FES half: dcv_linear_binning + statistics.compute_fes against the NumPy restatement of the same binned KDE
    (oracle/fes.py): node weights to 1e-9 (64-bit fixed-point accumulation, 2^-36 per contribution), FES to 1e-6 kJ/mol;
    identical grids from two runs (integer atomics commute)."""
    from deep_cartograph_amd import hip, statistics
    from oracle import fes as ofes

    rng = np.random.Generator(np.random.PCG64(11))
    X = np.concatenate([rng.normal(-0.4, 0.12, (40_000, d)), rng.normal(0.45, 0.2, (60_000, d))]).clip(-1.2, 1.2)
    lo, hi = [-1.0] * d, [1.0] * d
    P = torch.from_numpy(X).cuda()
    w1, out1 = hip.linear_binning(P, list(range(d)), lo, hi, bins)
    w2, _ = hip.linear_binning(P, list(range(d)), lo, hi, bins)
    assert torch.equal(w1, w2)
    exp = ofes.linear_binning(X, lo, hi, bins)
    assert out1 == int(np.sum(np.any((X < -1.0) | (X > 1.0), axis=1)))
    np.testing.assert_allclose(w1.cpu().numpy(), exp, atol=1e-9 * X.shape[0])
    fes, grid, bounds, err = statistics.compute_fes(X, 300.0, 0.05, bins, blocks=blocks, bounds=list(zip(lo, hi)))
    assert fes.shape == (bins,) * d and err.shape == fes.shape and abs(fes.min()) < 1e-12
    if blocks == 1:
        ref = ofes.binned_fes(X, 300.0, 0.05, bins, lo, hi)
        np.testing.assert_allclose(fes, ref, atol=1e-6)
    else:
        assert np.all(err >= 0) and np.isfinite(err).all()
