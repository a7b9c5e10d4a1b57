"""Size-independent properties of the HIP path at BASELINE.json's full single-GPU sizes (C3 5M x 256, C4 10M x 512 with
the bench batch, C5 20M x 4 points): shard additivity of every reduction a frame-sharded run all-reduces, exact scaling
laws, idempotence, chunking independence, and the data-parallel decomposition of a whole Deep-TICA step.  The oracle
cannot run at these sizes in seconds; it pins the same kernels at small sizes in the other test files.
Run on the GPU box: python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4_matrix():
    from deep_cartograph_amd.synth import synth_features

    X = synth_features(10_000_000, 512, k_slow=4, device="cuda")
    yield X
    del X
    torch.cuda.empty_cache()


def test_c4_statistics_and_normalisation_properties(c4_matrix):
    """10M x 512: column statistics are additive over frame shards (what a multi-GPU run all-reduces), scale exactly
    with a power-of-two factor, and the standardised matrix has mean 0 / std 1; standardising with (0, 1) is the identity."""
    from deep_cartograph_amd import hip

    X = c4_matrix
    n, F = X.shape
    raw = hip.col_stats_raw(X)
    cut = 3_333_337
    r1, r2 = hip.col_stats_raw(X[:cut]), hip.col_stats_raw(X[cut:])
    np.testing.assert_allclose((r1[:2] + r2[:2]).cpu().numpy(), raw[:2].cpu().numpy(), rtol=1e-12)
    assert torch.equal(torch.minimum(r1[2], r2[2]), raw[2]) and torch.equal(torch.maximum(r1[3], r2[3]), raw[3])
    X2 = X[:2_000_000] * 2.0                                   # exact in fp32: every sum doubles, every square quadruples
    a, b = hip.col_stats_raw(X[:2_000_000]), hip.col_stats_raw(X2)
    assert torch.equal(b[0], 2.0 * a[0]) and torch.equal(b[1], 4.0 * a[1]) and torch.equal(b[2:], 2.0 * a[2:])
    del X2
    st = hip.finalize_stats(raw, n)
    mean, std = torch.from_numpy(st["mean"]).cuda(), torch.from_numpy(st["std"]).cuda()
    Xn = hip.normalize(X[:4_000_000], mean, std)
    stn = hip.finalize_stats(hip.col_stats_raw(Xn), Xn.shape[0])
    st4 = hip.finalize_stats(hip.col_stats_raw(X[:4_000_000]), 4_000_000)
    exp_mean = (st4["mean"].astype(np.float64) - st["mean"]) / st["std"]       # the block's own mean in normalised units
    # fp32 subtraction of a mean up to 50 standard deviations from zero (scrambled columns): 1e-5 of a standard deviation
    np.testing.assert_allclose(stn["mean"], exp_mean, atol=1e-5)
    np.testing.assert_allclose(stn["std"], st4["std"].astype(np.float64) / st["std"], rtol=1e-5)
    again = hip.normalize(Xn, torch.zeros(F, device="cuda"), torch.ones(F, device="cuda"))
    assert torch.equal(again, Xn)


def test_c3_lagged_covariance_properties():
    """5M x 256, lag 10: the raw second-moment sums are additive over pair shards (with the lag-row halo), A is symmetric and its trace is the sum of squares the statistics kernel finds."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    n, F, lag = 5_000_000, 256, 10
    X = synth_features(n, F, k_slow=4, device="cuda")
    st = hip.finalize_stats(hip.col_stats_raw(X), n)
    hip.normalize(X, torch.from_numpy(st["mean"]).cuda(), torch.from_numpy(st["std"]).cuda(), out=X)
    P = n - lag
    raw = hip.lagged_cov_raw(X, P, lag).cpu().numpy()
    cut = 2_345_678                                             # pairs [0, cut) and [cut, P); the first shard borrows `lag` rows
    ra = hip.lagged_cov_raw(X[:cut + lag], cut, lag).cpu().numpy()
    rb = hip.lagged_cov_raw(X[cut:], P - cut, lag).cpu().numpy()
    A = raw[2 * F:2 * F + F * F].reshape(F, F)
    scale = np.sqrt(np.outer(np.diag(A), np.diag(A)))
    assert np.max(np.abs((ra + rb)[2 * F:2 * F + F * F].reshape(F, F) - A) / scale) < 2e-6
    assert np.max(np.abs((ra + rb)[2 * F + F * F:].reshape(F, F) - raw[2 * F + F * F:].reshape(F, F)) / scale) < 2e-6
    np.testing.assert_allclose((ra + rb)[:2 * F], raw[:2 * F], rtol=1e-9, atol=1e-6)
    assert np.max(np.abs(A - A.T) / scale) < 2e-6
    sumsq = hip.col_stats_raw(X[:P])[1].cpu().numpy()
    np.testing.assert_allclose(np.diag(A), sumsq, rtol=2e-5)
    raw0 = hip.lagged_cov_raw(X[:1_000_000], 1_000_000, 0).cpu().numpy()               # lag 0 (PCA): only A is formed
    np.testing.assert_allclose(np.diag(raw0[2 * F:2 * F + F * F].reshape(F, F)), hip.col_stats_raw(X[:1_000_000])[1].cpu().numpy(), rtol=2e-5)
    assert not raw0[2 * F + F * F:].any()
    del X
    torch.cuda.empty_cache()


def test_c3_tica_htica_calculators_fullsize(tmp_path):
    """The tica / htica CALCULATORS (cv_calculator.py:2249-2267, :2311-2384) at BASELINE C3's size, 5M x 256, lag 10 -- not only
    the covariance kernel under them.  Size-independent checks: the TICA weights solve the generalised eigenproblem of
    the covariances the kernel-level path gives for the same matrix (residual, C0-orthonormality, eigenvalues in
    descending order inside (0, 1)); they agree with the host solve of those kernel-level covariances; the
    projections are min-max normalised to [-1, 1]; hTICA (10 subspaces of 5 features) obeys the same equations inside the
    span it works in and its leading eigenvalue cannot exceed TICA's (a Rayleigh quotient over a subspace)."""
    from deep_cartograph_amd import hip, linalg
    from deep_cartograph_amd.cv_calculator import cv_calculators_map
    from deep_cartograph_amd.synth import synth_features

    n, F, lag, d = 5_000_000, 256, 10, 3
    X = synth_features(n, F, k_slow=4, device="cuda")
    cfg = {"dimension": d, "lag_time": lag, "features_normalization": "mean_std", "num_subspaces": 10, "subspaces_dimension": 5,
           "tica_regularization": 1e-6}
    out = {}
    for cv in ("tica", "htica"):
        calc = cv_calculators_map[cv](json_copy(cfg), str(tmp_path / cv))
        calc.set_training_matrix(X.clone())
        calc.create_output_folders()
        calc.compute_cv()
        calc.set_labels()
        calc.normalize_cv()
        proj = calc.project_data(calc.training_data, normalize_data=False)
        pr = proj.numpy() if not proj.is_cuda else proj.cpu().numpy()
        assert pr.shape == (n, d)
        np.testing.assert_allclose(pr.min(0), -1.0, atol=2e-5)
        np.testing.assert_allclose(pr.max(0), 1.0, atol=2e-5)
        out[cv] = (np.asarray(calc.cv, dtype=np.float64), calc)
    # the covariance-level path on the same standardised matrix
    calc = out["tica"][1]
    Xn = calc.training_data          # standardised in place by the linear calculators
    P = n - lag
    shift = torch.zeros(F, dtype=torch.float32, device="cuda")
    raw = hip.lagged_cov_raw(Xn, P, lag, shift).cpu().numpy()
    _, C0, Ct = hip.covariances_from_raw(raw, P, F)
    W = out["tica"][0]
    C0r = C0 + 1e-6 * np.eye(F)
    lam = np.diag(W.T @ Ct @ W) / np.diag(W.T @ C0r @ W)
    assert np.all(np.diff(lam) <= 1e-9) and np.all(lam > 0) and np.all(lam < 1)
    G = W.T @ C0 @ W
    np.testing.assert_allclose(G / np.sqrt(np.outer(np.diag(G), np.diag(G))), np.eye(d), atol=1e-5)   # C0-orthogonal directions
    resid = Ct @ W - C0r @ W * lam
    assert np.max(np.abs(resid)) < 2e-5 * np.max(np.abs(Ct @ W))
    ev, evec = linalg.tica_eigh(C0, Ct, 1e-6, d)
    np.testing.assert_allclose(lam, ev, rtol=1e-6)
    Wh = np.asarray(evec, dtype=np.float64)
    for k in range(d):   # same directions up to sign and scale as the host solve of the kernel-level covariances
        c = abs(W[:, k] @ C0 @ Wh[:, k]) / np.sqrt((W[:, k] @ C0 @ W[:, k]) * (Wh[:, k] @ C0 @ Wh[:, k]))
        assert abs(c - 1.0) < 1e-6, (k, c)
    Wt = out["htica"][0]
    lam_h = np.diag(Wt.T @ Ct @ Wt) / np.diag(Wt.T @ C0 @ Wt)
    assert np.all(lam_h > 0) and lam_h[0] <= lam[0] + 1e-6
    Gh = Wt.T @ C0 @ Wt
    np.testing.assert_allclose(Gh / np.sqrt(np.outer(np.diag(Gh), np.diag(Gh))), np.eye(d), atol=1e-4)
    del X
    torch.cuda.empty_cache()


def json_copy(x):
    import json

    return json.loads(json.dumps(x))


def test_c4_step_decomposes_over_two_shards(c4_matrix):
    """The bench's Deep-TICA step (524 208 pairs of the 10M x 512 matrix, MLP 512-256-128-4) decomposes the way the
    data-parallel run computes it: statistics of two half batches add up to the whole batch's, and with the summed
    statistics the two halves' gradients add up to the whole batch's gradient (tolerance 1e-4 of the largest entry:
    fp32 partial sums in a different order); the row-shared evaluation equals the two-half evaluation."""
    from deep_cartograph_amd import hip

    X = c4_matrix
    n, F = X.shape
    st = hip.finalize_stats(hip.col_stats_raw(X), n)
    Xn = hip.normalize(X[:1_200_000], torch.from_numpy(st["mean"]).cuda(), torch.from_numpy(st["std"]).cuda())
    dims, acts, lag, B = [F, 256, 128, 4], ["leaky_relu", "leaky_relu", None], 10, 524_208
    torch.manual_seed(43)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]
    params = [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins]
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=B, lag=lag, tica_reg=1e-6, lr=1e-3)
    eng.set_linears(params)
    eng.reset_log(64)
    sv, gv = eng.stats_view(), eng.grads_view()
    eng.forward(Xn, row0=0, batch=B)
    stats_full = sv.clone()
    eng.backward(Xn, row0=0, batch=B, global_batch=B, train=True)
    g_full = gv.clone()
    h = B // 2
    eng.forward(Xn, row0=0, batch=h)
    s1 = sv.clone()
    eng.forward(Xn, row0=h, batch=h)
    s2 = sv.clone()
    np.testing.assert_allclose((s1 + s2).cpu().numpy(), stats_full.cpu().numpy(), rtol=1e-9, atol=1e-6)
    sv.copy_(s1 + s2)
    eng.backward(Xn, row0=h, batch=h, global_batch=B, train=True)          # activations of the second half are resident
    g2 = gv.clone()
    eng.forward(Xn, row0=0, batch=h)
    sv.copy_(s1 + s2)
    eng.backward(Xn, row0=0, batch=h, global_batch=B, train=True)
    g1 = gv.clone()
    gs, gf = (g1 + g2).cpu().numpy(), g_full.cpu().numpy()
    w_count = sum(dims[i] * dims[i + 1] for i in range(2))                 # layers 0 and 1 (the last bias gradient is 0 + noise)
    assert np.max(np.abs(gs[:w_count] - gf[:w_count])) < 1e-4 * np.max(np.abs(gf[:w_count]))
    eng.set_row_sharing(False)
    eng.forward(Xn, row0=0, batch=B)
    np.testing.assert_allclose(sv.cpu().numpy(), stats_full.cpu().numpy(), rtol=1e-6, atol=1e-3)
    eng.close()


def test_c4_projection_chunking_and_linearity(c4_matrix):
    """Linear projection of 10M x 512 is linear in the weights and independent of how the frames are cut; the MLP
    inference of the engine gives the same rows whichever chunk they arrive in."""
    from deep_cartograph_amd import hip

    X = c4_matrix
    F = X.shape[1]
    g = torch.Generator(device="cuda").manual_seed(5)
    W1 = torch.randn(F, 4, device="cuda", generator=g) / 16
    W2 = torch.randn(F, 4, device="cuda", generator=g) / 16
    p1, _ = hip.project_linear(X, W1)
    p2, _ = hip.project_linear(X, W2)
    p12, mm = hip.project_linear(X, (W1 + W2).contiguous(), want_minmax=True)
    err = (p1 + p2 - p12).abs().max().item()
    assert err < 2e-4 * p12.abs().max().item()
    assert torch.equal(mm[0], p12.min(dim=0).values) and torch.equal(mm[1], p12.max(dim=0).values)
    pc, _ = hip.project_linear(X[3_000_001:7_000_003], (W1 + W2).contiguous())
    assert torch.equal(pc, p12[3_000_001:7_000_003])
    dims, acts = [F, 256, 128, 4], ["leaky_relu", "leaky_relu", None]
    torch.manual_seed(1)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=262_144, lag=0, tica_reg=1e-6)
    eng.set_linears([(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins])
    whole, _ = eng.infer(X[:2_000_000])
    part, _ = eng.infer(X[777_777:1_333_333])
    assert torch.equal(part, whole[777_777:1_333_333])
    eng.close()


def test_c5_kmeans_pass_properties():
    """20M x 4 points (CSV-rounded), k = 6: the Lloyd pass is additive over shards, counts every point once, is
    idempotent (a second pass with the same centres changes no label and reproduces the sums bit for bit), its inertia
    is the sum of the per-point distances it reports, and a Lloyd iteration does not increase the inertia."""
    from deep_cartograph_amd import hip

    n, d, k = 20_000_000, 4, 6
    g = torch.Generator(device="cuda").manual_seed(7)
    mu = torch.rand(k, d, device="cuda", generator=g, dtype=torch.float64) * 1.6 - 0.8
    P = mu[torch.randint(0, k, (n,), device="cuda", generator=g)] + 0.08 * torch.randn(n, d, device="cuda", generator=g, dtype=torch.float64)
    P = (P.clamp(-1, 1) * 1e4).round() / 1e4
    C = P[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
    lab = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    acc, md = hip.kmeans_step(P, C, lab, want_mindist=True)
    acc = acc.cpu().numpy()
    counts = acc[k * d:k * d + k]
    assert counts.sum() == n and acc[k * d + k + 1] == n            # every point counted once, every label changed from -1
    np.testing.assert_allclose(acc[k * d + k], md.sum().item(), rtol=1e-12)
    lab2 = lab.clone()
    acc2, _ = hip.kmeans_step(P, C, lab2)
    acc2 = acc2.cpu().numpy()
    assert acc2[k * d + k + 1] == 0 and torch.equal(lab, lab2)
    # the pass that also reports per-point distances keeps its points in registers, the plain pass streams them through the
    # per-wave LDS-DMA ring (kmeans.hip, round 4): two fixed summation orders -- equal counts, sums to rounding;
    # the SAME pass repeated is bit-identical
    np.testing.assert_array_equal(acc2[k * d:k * d + k], acc[k * d:k * d + k])
    np.testing.assert_allclose(acc2[:k * d + k + 1], acc[:k * d + k + 1], rtol=1e-13)
    acc2b, _ = hip.kmeans_step(P, C, lab2)
    np.testing.assert_array_equal(acc2b.cpu().numpy()[:k * d + k + 1], acc2[:k * d + k + 1])
    cut = 7_654_321
    la, lb = torch.full((cut,), -1, dtype=torch.int32, device="cuda"), torch.full((n - cut,), -1, dtype=torch.int32, device="cuda")
    a1, _ = hip.kmeans_step(P[:cut], C, la)
    a2, _ = hip.kmeans_step(P[cut:], C, lb)
    np.testing.assert_allclose((a1 + a2).cpu().numpy()[:k * d + k + 1], acc[:k * d + k + 1], rtol=1e-12)
    assert torch.equal(torch.cat([la, lb]), lab)
    newC = torch.from_numpy(acc[:k * d].reshape(k, d) / counts[:, None]).cuda()
    acc3, _ = hip.kmeans_step(P, newC, lab2)
    assert acc3.cpu().numpy()[k * d + k] <= acc[k * d + k]
    dist, rows = hip.nearest_rows(P, newC)
    assert bool(torch.all((rows >= 0) & (rows < n)))
    np.testing.assert_allclose(dist.cpu().numpy(), torch.linalg.norm(P[rows] - newC, dim=1).cpu().numpy(), rtol=1e-12, atol=1e-15)


def test_c5_deeptica_leg_properties():
    """C5 at full single-GPU size: 20M x 1024 float32 (82 GB resident), Deep-TICA 1024-256-128-4, lag 10.  The statistics
    and in-place standardisation of the whole matrix, training steps at F = 1024 (the loss of a held-out batch improves; the
    batch statistics of two half batches add up to the full batch's; row sharing equals the two-half evaluation), the
    projection of all 20M frames (chunking independence, extrema consistent with the output), then the '%.4f' seam and
    k-means with k-means++ seeding on the 20M x 4 projection (every point labelled once, seeding passes on the device)."""
    from deep_cartograph_amd import hip, statistics
    from deep_cartograph_amd.synth import synth_features

    n, F, lag, B = 20_000_000, 1024, 10, 65_526
    X = synth_features(n, F, k_slow=4, device="cuda")
    raw = hip.col_stats_raw(X)
    st = hip.finalize_stats(raw, n)
    hip.normalize(X, torch.from_numpy(st["mean"]).cuda(), torch.from_numpy(st["std"]).cuda(), out=X)
    stn = hip.finalize_stats(hip.col_stats_raw(X[:5_000_000]), 5_000_000)
    assert np.max(np.abs(stn["std"] - 1.0)) < 0.05 and np.max(np.abs(stn["mean"])) < 0.3    # a 5M block of a standardised AR(1) matrix
    dims, acts = [F, 256, 128, 4], ["leaky_relu", "leaky_relu", None]
    torch.manual_seed(43)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=B, lag=lag, tica_reg=1e-6, lr=1e-3)
    eng.set_linears([(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins])
    eng.reset_log(256)
    sv = eng.stats_view()
    eng.forward(X, row0=0, batch=B)
    full = sv.clone()
    h = B // 2
    eng.forward(X, row0=0, batch=h)
    s1 = sv.clone()
    eng.forward(X, row0=h, batch=h)
    np.testing.assert_allclose((s1 + sv).cpu().numpy(), full.cpu().numpy(), rtol=1e-9, atol=1e-6)
    eng.set_row_sharing(False)
    eng.forward(X, row0=0, batch=B)
    np.testing.assert_allclose(sv.cpu().numpy(), full.cpu().numpy(), rtol=1e-6, atol=1e-3)
    eng.set_row_sharing(True)
    held = 19_000_000
    eng.reset_log(256)
    eng.eval_step(X, row0=held, batch=B)
    for i in range(60):
        eng.train_step(X, row0=(i * B) % 16_000_000, batch=B)
    eng.eval_step(X, row0=held, batch=B)
    rec = eng.read_log()
    assert np.all(np.isfinite(rec[:, 0])) and rec[-1, 0] < rec[0, 0] - 0.05 and rec[-1, 0] >= -4.0   # -sum(eig^2) in [-d, 0]
    # projection of every frame: chunk independence and extrema
    out, mm = eng.infer(X, want_minmax=True)
    assert out.shape == (n, 4) and bool(torch.isfinite(out).all())
    assert torch.equal(mm[0], out.min(dim=0).values) and torch.equal(mm[1], out.max(dim=0).values)
    part, _ = eng.infer(X[7_777_777:8_123_456])
    assert torch.equal(part, out[7_777_777:8_123_456])
    eng.close()
    del X
    torch.cuda.empty_cache()
    # '%.4f' seam + k-means with k-means++ seeding on the 20M x 4 projection scaled to [-1, 1]
    cvs = ((out - (mm[1] + mm[0]) / 2) / ((mm[1] - mm[0]) / 2)).cpu().numpy().astype(np.float64)
    P = np.round(cvs, 4)
    labels, centers = statistics.kmeans_clustering(P, 6, 1)
    assert labels.shape == (n,) and set(np.unique(labels)) == set(range(6))
    assert centers.shape == (6, 4) and np.all(np.abs(centers) <= 1.0)
    # the centres are the member means of the last M-step; sklearn stops on the centre shift (tol) and re-labels once against
    # them, so they sit within the tolerance of the final members' means, and every point is nearest to its own centre
    for j in range(6):
        np.testing.assert_allclose(P[labels == j].mean(axis=0), centers[j], atol=5e-3)
    sub = P[::997]
    d2 = ((sub[:, None, :] - centers[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(np.argmin(d2, axis=1), labels[::997])


def test_c2_autoencoder_properties():
    """C2 at full size: 1M x 128, autoencoder 128-64-32-2-32-64-128, batch 4096.  The squared-error statistic of two half
    batches adds up to the full batch's, the reconstruction loss of a held-out batch falls over one epoch of training
    steps, an evaluation step leaves the parameters untouched, and the latent projection does not depend on chunking."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd.synth import synth_features

    n, F, B = 1_000_000, 128, 4096
    X = synth_features(n, F, k_slow=2, device="cuda")
    st = hip.finalize_stats(hip.col_stats_raw(X), n)
    Xn = hip.normalize(X, torch.from_numpy(st["mean"]).cuda(), torch.from_numpy(st["std"]).cuda())
    dims = [F, 64, 32, 2, 32, 64, F]
    acts = ["leaky_relu", "leaky_relu", None, "leaky_relu", "leaky_relu", None]
    torch.manual_seed(43)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(6)]
    eng = hip.Mlp("ae", dims, acts, max_batch=B, latent_layer=3, lr=1e-3)
    eng.set_linears([(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins])
    eng.set_feature_range(st["std"])
    eng.reset_log(512)
    sv = eng.stats_view()
    eng.forward(Xn, row0=0, batch=B)
    full = sv.clone()
    eng.forward(Xn, row0=0, batch=B // 2)
    s1 = sv.clone()
    eng.forward(Xn, row0=B // 2, batch=B // 2)
    np.testing.assert_allclose((s1 + sv).cpu().numpy(), full.cpu().numpy(), rtol=1e-10)
    held = 900_000
    before = eng.get_linears()
    eng.eval_step(Xn, row0=held, batch=B)
    after = eng.get_linears()
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(before, after))
    for i in range(195):   # one epoch of the 0.8 training share
        eng.train_step(Xn, row0=i * B, batch=B)
    eng.eval_step(Xn, row0=held, batch=B)
    rec = eng.read_log()
    assert np.all(np.isfinite(rec[:, 0])) and rec[-1, 0] < 0.8 * rec[0, 0]
    whole, mm = eng.infer(Xn, want_minmax=True)
    assert whole.shape == (n, 2)
    part, _ = eng.infer(Xn[123_457:345_679])
    assert torch.equal(part, whole[123_457:345_679])
    assert torch.equal(mm[0], whole.min(dim=0).values) and torch.equal(mm[1], whole.max(dim=0).values)
    eng.close()
