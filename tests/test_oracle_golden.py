"""Pin the CPU oracle against the golden vectors derived from the reference's own fixtures
(SURVEY.md section 8c).  CPU only."""
import numpy as np
import torch

from oracle import cluster as oc
from oracle import linear as ol
from oracle import nn as onn

CVS = ["pca", "tica", "htica", "ae", "deep_tica", "vae"]


def _normalized(features):
    X, _ = features
    st = ol.feature_stats(X)
    m, r = ol.prepare_normalization(st, "mean_std")
    return X, st, m, r, ol.normalize(X, m, r)


def test_stats_match_model_zip(features, golden_linear):
    X, st, m, r, Xn = _normalized(features)
    # means bit-exact, std within 1 ulp
    np.testing.assert_array_equal(m.astype(np.float32), golden_linear["pca.features_norm_mean"])
    np.testing.assert_allclose(r, golden_linear["pca.features_norm_range"], rtol=2e-7)


def test_pca_weights_and_projection(features, golden_linear, golden_proj):
    X, st, m, r, Xn = _normalized(features)
    W = ol.pca_cv(Xn, 2)
    np.testing.assert_allclose(W, golden_linear["pca.cv_weights"], atol=1e-6)
    cm, cr = ol.linear_cv_norm(Xn, W)
    P = ol.project_linear(Xn, W, cm, cr)
    np.testing.assert_array_equal(ol.csv_round4(P), golden_proj["pca"])


def test_model_zip_projection_exact(features, golden_linear, golden_proj):
    """a14/a16: the arrays of the bundled linear model.zip files project to the goldens."""
    X, _ = features
    for cv in ["pca", "tica", "htica"]:
        P = ol.project_linear(X, golden_linear[f"{cv}.cv_weights"], golden_linear[f"{cv}.cv_norm_mean"],
                              golden_linear[f"{cv}.cv_norm_range"], golden_linear[f"{cv}.features_norm_mean"],
                              golden_linear[f"{cv}.features_norm_range"])
        np.testing.assert_array_equal(ol.csv_round4(P), golden_proj[cv])


def test_tica_weights(features, golden_linear):
    """Tolerance 3e-4 abs: the fixture's own fp32 noise on this ill-conditioned 164x54 case
    (fp32 restatement 1.7e-4, fp64 7.9e-5)."""
    X, st, m, r, Xn = _normalized(features)
    W = ol.tica_cv(Xn, 1, 2)
    np.testing.assert_allclose(W, golden_linear["tica.cv_weights"], atol=3e-4)
    W64 = ol.tica_cv(Xn, 1, 2, dtype=torch.float64)
    np.testing.assert_allclose(W64, golden_linear["tica.cv_weights"], atol=1.5e-4)


def test_pairing_count(features, golden_linear):
    """Appendix A.3: only P = N - lag reproduces the fixture."""
    X, st, m, r, Xn = _normalized(features)
    for drop, ok in [(0, True), (1, False)]:
        x_t, x_lag = ol.timelagged_pairs(Xn, 1)
        if drop:
            x_t, x_lag = x_t[:-drop], x_lag[:-drop]
        _, ev, _ = ol.tica(x_t, x_lag, 2)
        close = np.allclose(ev.numpy(), golden_linear["tica.cv_weights"], atol=3e-4)
        assert close == ok


def test_htica_weights(features, golden_linear):
    X, st, m, r, Xn = _normalized(features)
    W = ol.htica_cv(Xn, 1, 2, 10, 5)
    np.testing.assert_allclose(W, golden_linear["htica.cv_weights"], atol=3e-4)


def _deep_tica_from_golden(g):
    m = onn.DeepTICAModel([54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None],
                          g["deep_tica.buffer.norm_in.mean"], g["deep_tica.buffer.norm_in.range"], 1e-6)
    sd = m.nn.state_dict()
    for i in (0, 3, 6):
        sd[f"{i}.weight"] = torch.from_numpy(g[f"deep_tica.param.nn.nn.{i}.weight"])
        sd[f"{i}.bias"] = torch.from_numpy(g[f"deep_tica.param.nn.nn.{i}.bias"])
    m.nn.load_state_dict(sd)
    m.eval()
    return m


def test_deep_tica_forward_and_buffers(features, golden_nn, golden_proj):
    X, _ = features
    g = golden_nn
    m = _deep_tica_from_golden(g)
    m.tica_evecs = torch.from_numpy(g["deep_tica.buffer.tica.evecs"])
    m.tica_mean = torch.from_numpy(g["deep_tica.buffer.tica.mean"])
    m.postprocessing = onn.Normalization(g["deep_tica.buffer.postprocessing.mean"], g["deep_tica.buffer.postprocessing.range"])
    with torch.no_grad():
        Y = m(torch.from_numpy(X)).numpy()
    np.testing.assert_allclose(Y, g["deep_tica.output"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(ol.csv_round4(g["deep_tica.output"]), golden_proj["deep_tica"])


def test_deep_tica_last_validation_batch_semantics(features, golden_nn):
    """Appendix A.6: seed 43 -> Linear inits -> randperm(163) -> 131/32 split; the exported
    TICA buffers are the TICA of the 32 validation pairs through the exported nn."""
    X, _ = features
    g = golden_nn
    m = _deep_tica_from_golden(g)
    gen = torch.manual_seed(43)
    onn.feed_forward([54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None])  # consume RNG
    tr, va = onn.split_indices(163, [0.8, 0.2], True, gen)
    assert (len(tr), len(va)) == (131, 32)
    Xt = torch.from_numpy(X)
    with torch.no_grad():
        f_t = m.forward_nn(Xt[:-1][va])
        f_lag = m.forward_nn(Xt[1:][va])
        ev, evecs, mu = onn.batch_tica(f_t, f_lag, 1e-6)
    np.testing.assert_allclose(evecs.numpy(), g["deep_tica.buffer.tica.evecs"], atol=2e-6)
    np.testing.assert_allclose(mu.numpy(), g["deep_tica.buffer.tica.mean"], atol=1e-7)
    # postprocessing = min-max over the x_t rows (a12)
    m.tica_evecs, m.tica_mean = evecs, mu
    onn.finalize_postprocessing(m, Xt[:-1])
    np.testing.assert_allclose(m.postprocessing.mean.numpy(), g["deep_tica.buffer.postprocessing.mean"], atol=1e-6)
    np.testing.assert_allclose(m.postprocessing.range.numpy(), g["deep_tica.buffer.postprocessing.range"], rtol=1e-5)
    assert onn.clamp_batch_size(256, 163, 0.8) == 128


def test_ae_forward(features, golden_nn, golden_proj):
    X, _ = features
    g = golden_nn
    m = onn.AEModel([54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None],
                    [2, 4, 8, 54], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None],
                    g["ae.buffer.norm_in.mean"], g["ae.buffer.norm_in.range"])
    for part in ("encoder", "decoder"):
        sd = getattr(m, part).state_dict()
        for i in (0, 3, 6):
            sd[f"{i}.weight"] = torch.from_numpy(g[f"ae.param.{part}.nn.{i}.weight"])
            sd[f"{i}.bias"] = torch.from_numpy(g[f"ae.param.{part}.nn.{i}.bias"])
        getattr(m, part).load_state_dict(sd)
    m.eval()
    onn.finalize_postprocessing(m, torch.from_numpy(X))
    np.testing.assert_allclose(m.postprocessing.mean.numpy(), g["ae.buffer.postprocessing.mean"], atol=1e-6)
    with torch.no_grad():
        Y = m(torch.from_numpy(X)).numpy()
    np.testing.assert_allclose(Y, g["ae.output"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(ol.csv_round4(g["ae.output"]), golden_proj["ae"])


def _reference_test_training(kind, X, m, r):
    """NonLinear.train of the reference's own test (tests/test_train_colvars.py:14-85): seed 42 + try 1, lengths
    [0.8, 0.2], batch 256 -> 128, max_epochs 1000, patience 20, min_delta 1e-5, Adam lr 1e-3, model_to_save 'last'."""
    kw = dict(seed_try=43, lengths=[0.8, 0.2], batch_size=onn.clamp_batch_size(256, 164, 0.8), shuffle=False, random_split=True,
              max_epochs=1000, check_val_every_n_epoch=1, save_check_every_n_epoch=1, patience=20, min_delta=1e-5,
              opt_kwargs={"lr": 1e-3, "weight_decay": 0}, model_to_save="last")
    Xt = torch.from_numpy(X)
    if kind == "ae":
        return onn.train(None, {"data": Xt}, build_model=lambda: onn.AEModel(
            [54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None], [2, 4, 8, 54], ["leaky_relu", "leaky_relu", None],
            [0.0, 0.0, None], m, r), **kw)
    return onn.train(None, {"data": Xt[:-1], "data_lag": Xt[1:]}, build_model=lambda: onn.DeepTICAModel(
        [54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None], m, r, 1e-6), **kw)


def test_training_reproduces_reference_models(features, golden_nn, golden_proj):
    """The training LOOP is pinned by the reference's fixtures: run on the reference's test configuration the oracle
    lands on the weights inside the reference's bundled ae_model.zip (544 epochs, 1088 Adam steps) and
    deep_tica_model.zip (21 epochs) -- DictLoader order, the 128 + 3 tail batch, Adam, early stopping and the 'last'
    checkpoint included.  Tolerances: float32 noise between torch builds (the fixtures were written by torch 2.1.2)."""
    X, _ = features
    g = golden_nn
    _, st, m, r, _ = _normalized(features)
    m, r = m.astype(np.float32), r.astype(np.float32)
    assert onn.clamp_batch_size(256, 164, 0.8) == 128
    # ---- autoencoder
    res = _reference_test_training("ae", X, m, r)
    assert len(res["metrics"]["epoch"]) == 544
    mod = res["model"]
    for part in ("encoder", "decoder"):
        sd = getattr(mod, part).state_dict()
        for i in (0, 3, 6):
            np.testing.assert_allclose(sd[f"{i}.weight"].numpy(), g[f"ae.param.{part}.nn.{i}.weight"], atol=5e-6)
            np.testing.assert_allclose(sd[f"{i}.bias"].numpy(), g[f"ae.param.{part}.nn.{i}.bias"], atol=5e-6)
    onn.finalize_postprocessing(mod, torch.from_numpy(X))
    with torch.no_grad():
        Y = mod(torch.from_numpy(X)).numpy()
    np.testing.assert_allclose(Y, g["ae.output"], atol=2e-5)
    assert np.mean(ol.csv_round4(Y) == golden_proj["ae"]) > 0.99
    # ---- Deep-TICA
    res = _reference_test_training("deep_tica", X, m, r)
    assert len(res["metrics"]["epoch"]) == 21
    mod = res["model"]
    sd = mod.nn.state_dict()
    for i in (0, 3, 6):
        np.testing.assert_allclose(sd[f"{i}.weight"].numpy(), g[f"deep_tica.param.nn.nn.{i}.weight"], atol=2e-5)
    for i in (0, 3):
        np.testing.assert_allclose(sd[f"{i}.bias"].numpy(), g[f"deep_tica.param.nn.nn.{i}.bias"], atol=1e-4)
    # the last bias is a pure offset the batch TICA removes: its exact gradient is zero, Adam turns the rounding noise
    # into +-lr steps (4e-3 away from the fixture after 42 steps) and tica.mean follows it -- their difference is pinned
    off = (sd["6.bias"].numpy() - mod.tica_mean.numpy()) - (g["deep_tica.param.nn.nn.6.bias"] - g["deep_tica.buffer.tica.mean"])
    assert np.max(np.abs(off)) < 2e-5
    np.testing.assert_allclose(mod.tica_evecs.numpy(), g["deep_tica.buffer.tica.evecs"], atol=5e-4)
    onn.finalize_postprocessing(mod, torch.from_numpy(X[:-1]))
    with torch.no_grad():
        Y = mod(torch.from_numpy(X)).numpy()
    # the exported TICA is that of the 32 validation pairs of the last epoch: it magnifies the 6e-6 weight noise to
    # 2e-4 in the eigenvectors and 5e-4 in the CV (torch 2.1.2 fixture vs the torch installed here)
    np.testing.assert_allclose(Y, g["deep_tica.output"], atol=1e-3)


# ----------------------------------------------------------------------------- clustering
def test_reference_cluster_goldens_reproduced(golden_cluster, golden_proj):
    """Appendix A.7: the reference module reproduces reference/traj_cluster/*.csv exactly."""
    for cv in CVS:
        np.testing.assert_array_equal(golden_cluster[f"{cv}.hier_labels"], golden_cluster[f"{cv}.golden_cluster"])
        np.testing.assert_array_equal(golden_cluster[f"{cv}.hier_centroid_flag"], golden_cluster[f"{cv}.golden_centroid"])
        rows = oc.find_centroid_rows(golden_proj[cv], golden_cluster[f"{cv}.hier_centroids"])
        flag = np.zeros(164, dtype=bool)
        flag[rows] = True
        np.testing.assert_array_equal(flag, golden_cluster[f"{cv}.golden_centroid"])


def test_kmeans_restatement_bit_exact_labels(golden_cluster, golden_proj):
    for cv in CVS:
        P = golden_proj[cv]
        for k in (3, 6):
            lab, cen, _, _ = oc.kmeans_restated(P, k, 5)
            np.testing.assert_array_equal(lab, golden_cluster[f"{cv}.kmeans_k{k}_labels"])
            np.testing.assert_allclose(cen, golden_cluster[f"{cv}.kmeans_k{k}_centroids"], atol=1e-12)
        lab, cen, _, _ = oc.kmeans_restated(P, 4, 1, initial_centroids=P[[0, 40, 80, 120]])
        np.testing.assert_array_equal(lab, golden_cluster[f"{cv}.kmeans_init_labels"])
    for tag in ("syn_a", "syn_b", "syn_c"):
        P = golden_cluster[f"{tag}.points"]
        lab, cen, _, _ = oc.kmeans_restated(P, 0, 1, initial_centroids=golden_cluster[f"{tag}.init"])
        np.testing.assert_array_equal(lab, golden_cluster[f"{tag}.init_labels"])
        np.testing.assert_allclose(cen, golden_cluster[f"{tag}.init_centroids"], atol=1e-12)
        k = golden_cluster[f"{tag}.init"].shape[0]
        lab, cen, _, _ = oc.kmeans_restated(P, k, 3)
        np.testing.assert_array_equal(lab, golden_cluster[f"{tag}.pp_labels"])
        rows = oc.find_centroid_rows(P, golden_cluster[f"{tag}.pp_centroids"])
        np.testing.assert_array_equal(np.sort(np.unique(rows)), golden_cluster[f"{tag}.pp_centroid_flag_rows"])


def test_kmeans_k_selection(golden_cluster, golden_proj):
    """optimize_clustering(kmeans) through the restatement picks the same k / labels."""
    for cv in ("pca", "ae"):
        P = golden_proj[cv]
        sets = [oc.kmeans_restated(P, k, 20)[0] for k in range(3, 11)]
        best = int(np.argmax(oc.combined_scores(P, sets)))
        np.testing.assert_array_equal(sets[best], golden_cluster[f"{cv}.kmeans_opt_labels"])


def test_binned_fes_approximates_the_exact_gaussian_kde():
    """f4: the binned KDE behind the FES (linear binning + grid convolution, what KDEpy's FFTKDE does) against the direct
    Gaussian sum on the same grid.  Linear binning widens the kernel by h^2 / 6 per axis (h = node spacing): 0.05 kJ/mol
    where the surface is below 10 kJ/mol, on nodes away from the boundary (mode='same' truncates the kernel there)."""
    from oracle import fes as ofes

    rng = np.random.Generator(np.random.PCG64(3))
    X = np.concatenate([rng.normal([-0.4, 0.2], 0.12, (3000, 2)), rng.normal([0.4, -0.3], 0.15, (3000, 2))]).clip(-0.95, 0.95)
    lo, hi = [-1.0, -1.0], [1.0, 1.0]
    bins = 201
    inner = np.abs(np.linspace(-1, 1, bins)) < 0.8
    for cols in ([0], [0, 1]):
        Xc = X[:, cols]
        a = ofes.binned_fes(Xc, 300.0, 0.05, bins, [lo[c] for c in cols], [hi[c] for c in cols])
        b = ofes.exact_kde_fes(Xc, 300.0, 0.05, bins, [lo[c] for c in cols], [hi[c] for c in cols])
        region = (b < 10.0) & (inner if len(cols) == 1 else inner[:, None] & inner[None, :])
        assert region.sum() > 20 and np.max(np.abs(a[region] - b[region])) < 5e-2
