"""Parity of the HIP MLP engine (Deep-TICA / autoencoder steps, training, inference) against
the torch-CPU autograd oracle.  Run on the GPU box: python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest
import torch

from oracle import linear as ol
from oracle import nn as onn

pytestmark = pytest.mark.gpu


def ar_features(n, F, seed, k_slow=3):
    """SURVEY.md section 8d generator (small): AR(1) slow modes mixed into F features."""
    rng = np.random.Generator(np.random.PCG64(seed))
    T = np.array([2000, 700, 250, 90][:k_slow], dtype=np.float64)
    rho = np.exp(-1.0 / T)
    z = np.zeros((n, k_slow))
    eta = rng.standard_normal((n, k_slow))
    for t in range(1, n):
        z[t] = rho * z[t - 1] + np.sqrt(1 - rho ** 2) * eta[t]
    A = rng.standard_normal((F, k_slow)) / np.sqrt(k_slow)
    X = z @ A.T + 0.5 * rng.standard_normal((n, F))
    X = X * rng.uniform(0.1, 10, F) + rng.uniform(-5, 5, F)
    return X.astype(np.float32)


def normalized(X):
    st = ol.feature_stats(X)
    m, r = ol.prepare_normalization(st, "mean_std")
    return ol.normalize(X, m, r), m.astype(np.float32), r.astype(np.float32)


def linears_of(seq):
    return [m for m in seq if isinstance(m, torch.nn.Linear)]


def push_params(eng, lins):
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins])


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("mode", ["split", "native"])
@pytest.mark.parametrize("dims,n,lag,batch,gather", [
    ([54, 16, 8, 2], 164, 1, 131, True),
    ([54, 16, 8, 2], 164, 1, 100, False),
    ([256, 128, 64, 4], 6000, 10, 2048, True),
    ([64, 32, 3], 3000, 5, 777, False),
    ([40, 5], 1000, 2, 500, True),
    # the contract batch: 8192 pairs + lag = 8202 shared rows, one ragged row tile past a multiple of the
    # CU count -- the products take the contraction-split tail tile (gemm.h: GemmDims::tail_split).  tanh layers: with a
    # million leaky-ReLU units per batch a few pre-activations sit within float32 rounding of the kink, and one flipped
    # slope (1 vs 0.01) moves a gradient entry by 1e-4 of the largest one -- in the float32 oracle just as in either
    # arithmetic flavour of the engine (measured: FP32-input MFMA 3.8e-4, split 3e-7 on this very batch)
    ([512, 256, 128, 3, "tanh"], 8300, 10, 8192, False),
    # 4096 pairs + lag = 4106 shared rows: the row-limited weight-gradient plan (mlp.hip: wgrad_plan) takes 16 contraction
    # chunks of 257 rows -- chunk ends that are no stage multiple (the stage tail goes through registers), 64 x 64 tiles
    ([512, 256, 128, 3, "tanh"], 4200, 10, 4096, False),
    # the reference's DEFAULT loader at the contract batch (random split + shuffle: bench.py's `shuffled` block): 8192
    # gathered pairs = 2 x 8192 rows through the int64 index, the grouped weight / input gradient launch of mixed tile
    # families (pair.hip), 64 split-K slabs for layer 0
    ([512, 256, 128, 3, "tanh"], 8400, 10, 8192, True),
])
def test_deeptica_step_matches_autograd(features, dims, n, lag, batch, gather, mode):
    """One Deep-TICA step (statistics, loss, every gradient) against a FLOAT64 run of the autograd oracle on the same
    float32 parameters and inputs, in both arithmetic flavours of the matrix products.  Tolerance 2e-5 of the
    largest gradient entry per tensor (the engine computes in float32; the float32 oracle itself sits 1e-6 .. 3e-3
    from the float64 one on these cases -- its d x d Cholesky / eigh run in float32)."""
    import copy

    from deep_cartograph_amd import hip

    hidden_act = "leaky_relu"
    if isinstance(dims[-1], str):
        dims, hidden_act = dims[:-1], dims[-1]
    X = features[0] if dims[0] == 54 else ar_features(n, dims[0], 11)
    Xn, _, _ = normalized(X)
    P = Xn.shape[0] - lag
    acts = [hidden_act] * (len(dims) - 2) + [None]
    torch.manual_seed(3)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    ref64 = copy.deepcopy(ref).double()
    prev = hip.get_gemm_mode()
    hip.set_gemm_mode(mode)
    try:
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
        lins = linears_of(ref64.nn)
        push_params(eng, linears_of(ref.nn))
        Xd = torch.from_numpy(Xn).cuda()
        if gather:
            idx = torch.randperm(P)[:batch].contiguous()
            kw = dict(idx=idx.cuda())
        else:
            idx = torch.arange(7, 7 + batch)
            kw = dict(row0=7, batch=batch)
        eng.reset_log(2)
        eng.forward(Xd, **kw)
        stats = eng.stats_view().cpu().numpy()
        eng.backward(Xd, **kw)
        g = eng.grads_view().cpu().numpy()
        rec = eng.read_log()[0]
        eng.close()
    finally:
        hip.set_gemm_mode(prev)
    xt = torch.from_numpy(Xn).double()
    loss, _ = ref64.step(xt[idx], xt[idx + lag])
    loss.backward()
    with torch.no_grad():
        f_t = ref64.forward_nn(xt[idx])
        f_l = ref64.forward_nn(xt[idx + lag])
    d = dims[-1]
    np.testing.assert_allclose(stats[:d], f_t.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[2 * d:2 * d + d * d].reshape(d, d), (f_t.T @ f_t).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(stats[2 * d + d * d:].reshape(d, d), (f_t.T @ f_l).numpy(), rtol=2e-5, atol=2e-5)
    assert abs(rec[0] - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert rec[1] == batch
    worst = 0.0
    for l, lin in enumerate(lins):
        wo, bo = eng.offsets[l]
        gw = lin.weight.grad.numpy()
        gb = lin.bias.grad.numpy()
        ew = rel_err(g[wo:wo + gw.size].reshape(gw.shape), gw)
        worst = max(worst, ew)
        assert ew < 2e-5, f"layer {l} weight: {ew:.2e}"
        if l < len(lins) - 1:
            eb = rel_err(g[bo:bo + gb.size], gb)
            worst = max(worst, eb)
            assert eb < 2e-5, f"layer {l} bias: {eb:.2e}"
        else:
            # the loss is invariant to a constant shift of the outputs (TICA removes the mean), so
            # the exact gradient of the last bias is 0: the engine holds float32 rounding noise only
            assert np.max(np.abs(g[bo:bo + gb.size])) < 2e-5 * max(1.0, np.max(np.abs(gw))), f"layer {l} bias"
    print(f"{mode} {dims} batch {batch}: worst gradient deviation from float64 = {worst:.2e} of the largest entry")


@pytest.mark.parametrize("mode", ["split", "native"])
def test_contract_batch_leaky_relu_step_kink_aware(mode):
    """The bench's own configuration -- 512-256-128-4, leaky-ReLU, 8192 contiguous pairs + lag 10 -- against float64.
    Among the three million hidden units of such a batch a few pre-activations sit within float32 rounding of the kink,
    and one flipped slope (1 vs 0.01) moves a gradient entry by 1e-4 of the largest one in ANY float32 implementation.
    Kink-aware criterion: (i) the engine's slope decisions (sign of its stored activations) may differ from the float64
    ones only where the float64 pre-activation is within 1e-5 of zero; (ii) given the engine's slope pattern -- the
    network is then piecewise linear with that pattern fixed -- every gradient tensor, the statistics and the loss must
    agree with the float64 evaluation to the tolerances of test_deeptica_step_matches_autograd."""
    from deep_cartograph_amd import hip

    dims, lag, batch, row0 = [512, 256, 128, 4], 10, 8192, 7
    acts = ["leaky_relu", "leaky_relu", None]
    Xn, _, _ = normalized(ar_features(8300, dims[0], 31))
    torch.manual_seed(6)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    prev = hip.get_gemm_mode()
    hip.set_gemm_mode(mode)
    R = batch + lag
    try:
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
        push_params(eng, linears_of(ref.nn))
        Xd = torch.from_numpy(Xn).cuda()
        eng.reset_log(2)
        eng.forward(Xd, row0=row0, batch=batch)
        stats = eng.stats_view().cpu().numpy()
        H = [eng.layer_output(l, R).cpu().numpy() for l in (0, 1)]
        eng.backward(Xd, row0=row0, batch=batch)
        g = eng.grads_view().cpu().numpy()
        rec = eng.read_log()[0]
        offsets = list(eng.offsets)
        eng.close()
    finally:
        hip.set_gemm_mode(prev)
    W = [l.weight.detach().double().clone().requires_grad_(True) for l in linears_of(ref.nn)]
    b = [l.bias.detach().double().clone().requires_grad_(True) for l in linears_of(ref.nn)]
    x = torch.from_numpy(Xn[row0:row0 + R]).double()
    h, flips = x, []
    for l in (0, 1):
        z = h @ W[l].T + b[l]
        m_eng = torch.from_numpy(H[l] > 0)
        differ = m_eng != (z.detach() > 0)
        flips.append(int(differ.sum()))
        if differ.any():   # (i): only units within float32 rounding of the kink
            assert float(z.detach().abs()[differ].max()) < 1e-5, f"layer {l}: slope decision differs away from the kink"
        h = z * torch.where(m_eng, 1.0, 0.01).double()   # (ii): the engine's slope pattern, fixed
    f = h @ W[2].T + b[2]
    f_t, f_l = f[:batch], f[lag:lag + batch]
    evals, _, _ = onn.batch_tica(f_t, f_l, 1e-6)
    loss = onn.deeptica_loss(evals)
    loss.backward()
    d = dims[-1]
    np.testing.assert_allclose(stats[:d], f_t.detach().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[2 * d:2 * d + d * d].reshape(d, d), (f_t.T @ f_t).detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(stats[2 * d + d * d:].reshape(d, d), (f_t.T @ f_l).detach().numpy(), rtol=2e-5, atol=2e-5)
    assert abs(rec[0] - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    worst = 0.0
    for l in range(3):
        wo, bo = offsets[l]
        gw, gb = W[l].grad.numpy(), b[l].grad.numpy()
        ew = rel_err(g[wo:wo + gw.size].reshape(gw.shape), gw)
        worst = max(worst, ew)
        assert ew < 2e-5, f"layer {l} weight: {ew:.2e}"
        if l < 2:
            eb = rel_err(g[bo:bo + gb.size], gb)
            worst = max(worst, eb)
            assert eb < 2e-5, f"layer {l} bias: {eb:.2e}"
        else:
            assert np.max(np.abs(g[bo:bo + gb.size])) < 2e-5 * max(1.0, np.max(np.abs(gw))), "last bias"
    print(f"{mode} contract batch, leaky-ReLU: slope decisions differing from float64 (all within 1e-5 of the kink): {flips}; "
          f"worst gradient deviation from float64 given the slope pattern = {worst:.2e} of the largest entry")


def test_contract_batch_step_is_reproducible():
    """The contraction-split tail tile (8202 rows: 512 regular + 32 tail-chunk workgroups per product) adds its chunks up in
    chunk order whatever the arrival order: statistics, loss and every gradient of repeated steps on the same weights are
    bit-identical, in two engines and across repetitions."""
    from deep_cartograph_amd import hip

    dims, lag, batch = [512, 256, 128, 3], 10, 8192
    acts = ["leaky_relu", "leaky_relu", None]
    Xn, _, _ = normalized(ar_features(8300, dims[0], 21))
    torch.manual_seed(4)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    Xd = torch.from_numpy(Xn).cuda()
    seen = []
    for _ in range(2):
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
        push_params(eng, linears_of(ref.nn))
        eng.reset_log(8)
        for _rep in range(3):
            eng.forward(Xd, row0=5, batch=batch)
            stats = eng.stats_view().cpu().numpy().copy()
            eng.backward(Xd, row0=5, batch=batch)
            seen.append((stats, eng.grads_view().cpu().numpy().copy()))
        log = eng.read_log()
        assert np.all(log[:, 0] == log[0, 0])
        eng.close()
    for st, g in seen[1:]:
        np.testing.assert_array_equal(st, seen[0][0])
        np.testing.assert_array_equal(g, seen[0][1])


def test_deeptica_row_sharing_equivalence():
    """Contiguous batches evaluate the network once on the batch + lag rows both halves share; the
    result must equal the two-halves evaluation (same rows through the same weights)."""
    from deep_cartograph_amd import hip

    n, F, lag, batch = 5000, 96, 7, 1500
    dims, acts = [F, 48, 24, 3], ["leaky_relu", "tanh", None]
    Xn, _, _ = normalized(ar_features(n, F, 9))
    torch.manual_seed(5)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    Xd = torch.from_numpy(Xn).cuda()
    out = {}
    for share in (True, False):
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
        push_params(eng, linears_of(ref.nn))
        eng.set_row_sharing(share)
        eng.reset_log(2)
        eng.forward(Xd, row0=11, batch=batch)
        stats = eng.stats_view().cpu().numpy().copy()
        eng.backward(Xd, row0=11, batch=batch)
        out[share] = (stats, eng.read_log()[0].copy(), eng.grads_view().cpu().numpy().copy())
        eng.close()
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-12)   # identical outputs -> identical float64 sums
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=1e-10)
    g1, g0 = out[True][2], out[False][2]
    assert np.max(np.abs(g1 - g0)) < 2e-6 * np.max(np.abs(g0))           # same gradient, different fp32 summation order
    # and a gathered batch with the same samples agrees too
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
    push_params(eng, linears_of(ref.nn))
    eng.reset_log(2)
    idx = torch.arange(11, 11 + batch).cuda()
    eng.forward(Xd, idx=idx)
    eng.backward(Xd, idx=idx)
    np.testing.assert_allclose(eng.read_log()[0], out[True][1], rtol=1e-10)
    eng.close()


# measured after 30 Adam steps: loss 4.6e-6 (relative), weights 1.3e-7, hidden biases 7.6e-8.  The other side is the FLOAT32 oracle,
# whose own rounding depends on the host BLAS of the box it runs on: ~7 x the measured deviations rather than 3 x
DT_TRAIN_TOL = {"loss": 3e-5, "w": 1e-6, "b": 6e-7}


def test_deeptica_training_matches_oracle():
    from deep_cartograph_amd import hip

    n, F, lag, bs = 3000, 32, 5, 256
    dims, acts = [F, 16, 8, 2], ["leaky_relu", "leaky_relu", None]
    Xn, _, _ = normalized(ar_features(n, F, 5))
    P = n - lag
    gen = torch.manual_seed(44)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    tr, va = onn.split_indices(P, [0.8, 0.2], True, gen)
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=bs, lag=lag, tica_reg=1e-6, lr=1e-3)
    lins = linears_of(ref.nn)
    push_params(eng, lins)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    xt = torch.from_numpy(Xn)
    Xd = xt.cuda()
    ref_losses = []
    eng.reset_log(256)
    for epoch in range(3):
        for b in onn.batches(tr, bs, False):
            opt.zero_grad()
            loss, _ = ref.step(xt[b], xt[b + lag])
            loss.backward()
            opt.step()
            ref_losses.append(float(loss))
            eng.train_step(Xd, idx=b.cuda())
    log = eng.read_log()
    assert log.shape[0] == len(ref_losses)
    dev_loss = float(np.max(np.abs(log[:, 0] - np.asarray(ref_losses)) / np.maximum(np.abs(ref_losses), 0.1)))
    dev_w = max(float(np.max(np.abs(w - lin.weight.detach().numpy()))) for (w, _), lin in zip(eng.get_linears(), lins))
    dev_b = max(float(np.max(np.abs(b - lin.bias.detach().numpy()))) for (_, b), lin in zip(eng.get_linears()[:-1], lins[:-1]))
    print(f"deep-tica training vs float32 oracle after {len(ref_losses)} steps: loss {dev_loss:.2e} (relative), weights {dev_w:.2e}, hidden biases {dev_b:.2e}")
    np.testing.assert_allclose(log[:, 0], ref_losses, rtol=DT_TRAIN_TOL["loss"], atol=DT_TRAIN_TOL["loss"] * 0.1)
    for l, ((w, b), lin) in enumerate(zip(eng.get_linears(), lins)):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=DT_TRAIN_TOL["w"])
        if l < len(lins) - 1:  # last bias: zero exact gradient, Adam amplifies rounding noise on both sides
            np.testing.assert_allclose(b, lin.bias.detach().numpy(), atol=DT_TRAIN_TOL["b"])
    # validation pass: eval steps log the loss and leave the parameters alone
    before = eng.get_linears()
    eng.reset_log(8)
    for b in onn.batches(va, bs, False):
        eng.eval_step(Xd, idx=b.cuda())
    vlog = eng.read_log()
    ref.eval()
    with torch.no_grad():
        vref = [float(ref.step(xt[b], xt[b + lag])[0]) for b in onn.batches(va, bs, False)]
    np.testing.assert_allclose(vlog[:, 0], vref, rtol=2e-3, atol=2e-4)
    for (w0, b0), (w1, b1) in zip(before, eng.get_linears()):
        np.testing.assert_array_equal(w0, w1)
    # logged C0 / Ctau reproduce the oracle's batch TICA eigenvalues
    d = 2
    rec = vlog[-1]
    C0 = torch.tensor(rec[2:2 + d * d].reshape(d, d))
    Ct = torch.tensor(rec[2 + d * d:2 + 2 * d * d].reshape(d, d))
    ev, _ = ol.cholesky_eigh(Ct, C0, 1e-6)
    with torch.no_grad():
        b = onn.batches(va, bs, False)[-1]
        ev_ref, _, _ = onn.batch_tica(ref.forward_nn(xt[b]), ref.forward_nn(xt[b + lag]), 1e-6)
    np.testing.assert_allclose(ev.numpy(), ev_ref.numpy(), atol=5e-4)
    eng.close()


def test_ae_training_matches_oracle(features):
    from deep_cartograph_amd import hip

    X = features[0]
    Xn, m, r = normalized(X)
    F = 54
    gen = torch.manual_seed(45)
    ref = onn.AEModel([F, 16, 8, 2], ["leaky_relu", "leaky_relu", None], None, [2, 4, 8, F], ["leaky_relu", "leaky_relu", None], None, m, r)
    tr, va = onn.split_indices(164, [0.8, 0.2], True, gen)
    dims = [F, 16, 8, 2, 4, 8, F]
    acts = ["leaky_relu", "leaky_relu", None, "leaky_relu", "leaky_relu", None]
    eng = hip.Mlp("ae", dims, acts, max_batch=64, latent_layer=3, lr=1e-3)
    lins = linears_of(ref.encoder) + linears_of(ref.decoder)
    push_params(eng, lins)
    eng.set_feature_range(r)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    xt = torch.from_numpy(X)
    Xd = torch.from_numpy(Xn).cuda()
    ref_losses = []
    eng.reset_log(64)
    # first step: gradients
    b0 = onn.batches(tr, 64, False)[0]
    eng.forward(Xd, idx=b0.cuda())
    eng.backward(Xd, idx=b0.cuda())
    loss, _ = ref.step(xt[b0])
    opt.zero_grad()
    loss.backward()
    g = eng.grads_view().cpu().numpy()
    for l, lin in enumerate(lins):
        wo, bo = eng.offsets[l]
        gw = lin.weight.grad.numpy()
        assert rel_err(g[wo:wo + gw.size].reshape(gw.shape), gw) < 1e-3, f"layer {l}"
        assert rel_err(g[bo:bo + lin.bias.numel()], lin.bias.grad.numpy()) < 1e-3, f"bias {l}"
    eng.reset_log(64)
    for epoch in range(5):
        for b in onn.batches(tr, 64, False):
            opt.zero_grad()
            loss, _ = ref.step(xt[b])
            loss.backward()
            opt.step()
            ref_losses.append(float(loss))
            eng.train_step(Xd, idx=b.cuda())
    log = eng.read_log()
    np.testing.assert_allclose(log[:, 0], ref_losses, rtol=1e-4)
    for (w, b), lin in zip(eng.get_linears(), lins):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=1e-4)
    # encoder inference + min/max
    ref.eval()
    with torch.no_grad():
        Y = ref.forward_cv(xt).numpy()
    out, mm = eng.infer(Xd, want_minmax=True)
    np.testing.assert_allclose(out.cpu().numpy(), Y, atol=2e-5)
    np.testing.assert_allclose(mm.cpu().numpy(), np.stack([Y.min(0), Y.max(0)]), atol=2e-5)
    eng.close()


@pytest.mark.parametrize("model", ["ae", "deep_tica"])
def test_ragged_width_training_matches_oracle(model):
    """Widths that are no multiple of 4 (odd weight counts, rows that are not 16-byte aligned): the scalar loaders of the
    block engine, the scalar items of the gradient reduction, the partial units of the fused small-network staging and
    its 4-byte partial stores.  Same step sequence as the oracle (torch CPU autograd, float32), SGD with momentum so
    that a parameter with an exactly-zero gradient is not noise-driven."""
    from deep_cartograph_amd import hip

    n, F, lag, bs = 1500, 37, 3, 200
    X = ar_features(n, F, 11)
    Xn, m, r = normalized(X)
    torch.manual_seed(46)
    if model == "ae":
        ref = onn.AEModel([F, 19, 7, 3], ["leaky_relu", "tanh", None], None, [3, 7, 19, F], ["leaky_relu", "tanh", None], None, m, r)
        dims, acts = [F, 19, 7, 3, 7, 19, F], ["leaky_relu", "tanh", None, "leaky_relu", "tanh", None]
        eng = hip.Mlp("ae", dims, acts, max_batch=bs, latent_layer=3, optimizer="SGD", lr=1e-2, momentum=0.9)
        lins = linears_of(ref.encoder) + linears_of(ref.decoder)
        xt = torch.from_numpy(X)
    else:
        dims, acts = [F, 19, 7, 3], ["leaky_relu", "tanh", None]
        ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=bs, lag=lag, tica_reg=1e-6, optimizer="SGD", lr=1e-2, momentum=0.9)
        lins = linears_of(ref.nn)
        xt = torch.from_numpy(Xn)
    push_params(eng, lins)
    if model == "ae":
        eng.set_feature_range(r)
    opt = torch.optim.SGD(ref.parameters(), lr=1e-2, momentum=0.9)
    Xd = torch.from_numpy(Xn).cuda()
    eng.reset_log(64)
    ref_losses = []
    for step in range(12):
        b = torch.arange(step * 97, step * 97 + (bs if step % 3 else bs - 13))   # ragged batch every third step
        opt.zero_grad()
        loss = ref.step(xt[b])[0] if model == "ae" else ref.step(xt[b], xt[b + lag])[0]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
        if step % 2:
            eng.train_step(Xd, idx=b.cuda())                 # gathered rows
        else:
            eng.train_step(Xd, row0=int(b[0]), batch=len(b))  # contiguous rows
    log = eng.read_log()
    dev_w = max(float(np.max(np.abs(w - lin.weight.detach().numpy()))) for (w, _), lin in zip(eng.get_linears(), lins))
    dev_b = max(float(np.max(np.abs(b_ - lin.bias.detach().numpy()))) for (_, b_), lin in zip(eng.get_linears(), lins))
    print(f"ragged widths, {model}: loss dev {np.max(np.abs(log[:, 0] - ref_losses) / np.abs(ref_losses)):.2e}, weights {dev_w:.2e}, biases {dev_b:.2e}")
    # measured: loss 1.3e-7 (ae) / 2.0e-6 (deep_tica), weights 1.4e-7, biases 1.3e-7
    np.testing.assert_allclose(log[:, 0], ref_losses, rtol=8e-6)
    assert dev_w < 5e-7 and dev_b < 5e-7
    eng.close()


def test_infer_reproduces_reference_torchscript(features, golden_nn, golden_proj):
    """a15: the bundled deep_tica model.zip (parameters + buffers) through the HIP engine
    gives the reference's own outputs."""
    from deep_cartograph_amd import hip

    X = features[0]
    g = golden_nn
    mean = g["deep_tica.buffer.norm_in.mean"]
    rng = g["deep_tica.buffer.norm_in.range"]
    Xd = torch.from_numpy(X).cuda()
    Xn = hip.normalize(Xd, torch.from_numpy(mean).cuda(), torch.from_numpy(rng).cuda())
    eng = hip.Mlp("deep_tica", [54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], max_batch=100, lag=1)
    eng.set_linears([(g[f"deep_tica.param.nn.nn.{i}.weight"], g[f"deep_tica.param.nn.nn.{i}.bias"]) for i in (0, 3, 6)])
    out, _ = eng.infer(Xn, tmean=torch.from_numpy(g["deep_tica.buffer.tica.mean"]).cuda(),
                       tevecs=torch.from_numpy(g["deep_tica.buffer.tica.evecs"]).cuda(),
                       pmean=torch.from_numpy(g["deep_tica.buffer.postprocessing.mean"]).cuda(),
                       prange=torch.from_numpy(g["deep_tica.buffer.postprocessing.range"]).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g["deep_tica.output"], atol=2e-5)
    assert np.mean(ol.csv_round4(out.cpu().numpy()) == golden_proj["deep_tica"]) > 0.97
    eng.close()


@pytest.mark.parametrize("dims,acts,n,cap", [
    ([70, 33, 17, 3], ["tanh", "leaky_relu", None], 1000, 384),       # ragged widths, chunked (3 calls)
    ([512, 256, 128, 4], ["leaky_relu", "leaky_relu", None], 5000, 8192),
    ([54, 5], [None], 164, 164),                                      # a single Linear
])
def test_input_sensitivity_matches_autograd(dims, acts, n, cap):
    """dcv_mlp_input_sensitivity: sum_r |d(g . net(xn_r))/d xn_i| * scale_i against float64
    autograd through the same Linear chain; tolerance 2e-4 relative (fp32 products vs float64)."""
    from deep_cartograph_amd import hip

    rng = np.random.default_rng(5)
    Xn = rng.standard_normal((n, dims[0])).astype(np.float32)
    torch.manual_seed(9)
    seq = onn.feed_forward(dims, acts)
    lins = linears_of(seq)
    g = rng.standard_normal(dims[-1]).astype(np.float32)
    scale = rng.uniform(0.5, 2.0, dims[0]).astype(np.float32)
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=cap, lag=0, tica_reg=1e-6)
    push_params(eng, lins)
    got = eng.input_sensitivity(torch.from_numpy(Xn).cuda(), torch.from_numpy(g).cuda(), torch.from_numpy(scale).cuda()).cpu().numpy()
    ref = seq.double()
    x = torch.from_numpy(Xn).double().requires_grad_(True)
    out = (ref(x) * torch.from_numpy(g).double()).sum()
    grad = torch.autograd.grad(out, x)[0].numpy()
    exp = (np.abs(grad) * scale.astype(np.float64)).sum(axis=0)
    assert got.shape == exp.shape
    np.testing.assert_allclose(got, exp, rtol=2e-4)
    eng.close()


# ----------------------------------------------------------------------------- two ranks on one GPU (gloo)
def _dp_rank(rank, world, port, tmpdir):
    """One data-parallel rank of the bench's collective path: forward, all-reduce of the batch statistics, backward,
    all-reduce of the gradient buffer, Adam -- on this rank's contiguous block of frames."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deep_cartograph_amd import hip

    torch.cuda.set_device(0)
    z = np.load(os.path.join(tmpdir, "dp.npz"))
    Xn, dims, lag, lb, steps = z["Xn"], [int(v) for v in z["dims"]], int(z["lag"]), int(z["lb"]), int(z["steps"])
    n_local = Xn.shape[0] // world
    Xd = torch.from_numpy(Xn[rank * n_local:(rank + 1) * n_local]).cuda()
    acts = ["leaky_relu"] * (len(dims) - 2) + [None]
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=lb, lag=lag, tica_reg=1e-6, lr=1e-3)
    eng.set_linears([(z[f"w{i}"], z[f"b{i}"]) for i in range(len(dims) - 1)])
    eng.reset_log(4 * steps + 8)
    sv, gv = eng.stats_view(), eng.grads_view()
    for i in range(steps):
        r0 = i * lb
        if i % 2 == 0:   # the engine's own sequence, in both forms: the two-piece one (upper-layer gradients all-reduced
            # under the layer-0 weight gradient; the default from 32768 rows per rank up) and the one-piece one
            os.environ["DCV_DP_OVERLAP"] = "1" if i % 4 == 0 else "0"
            eng.data_parallel_step(Xd, dist, lb * world, row0=r0, batch=lb)
            del os.environ["DCV_DP_OVERLAP"]
        else:            # the same step written out with one gradient all-reduce
            eng.forward(Xd, row0=r0, batch=lb)
            dist.all_reduce(sv, op=dist.ReduceOp.SUM)
            eng.backward(Xd, row0=r0, batch=lb, global_batch=lb * world, train=True)
            dist.all_reduce(gv, op=dist.ReduceOp.SUM)
            eng.apply()
    torch.cuda.synchronize()
    lin = eng.get_linears()
    np.savez(os.path.join(tmpdir, f"dp_rank{rank}.npz"), loss=eng.read_log()[:steps, 0], **{f"w{i}": w for i, (w, _) in enumerate(lin)})
    eng.close()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_match_single_process(tmp_path):
    """SURVEY 8e, Deep-TICA step: two ranks (frame blocks, batch statistics and gradients all-reduced) follow the same
    trajectory as one process given the union batch through a gather index; tolerance 2e-5 on the weights after 4 steps
    (summation order of the fp32 partial sums differs), losses 1e-5."""
    import socket

    import torch.multiprocessing as mp

    from deep_cartograph_amd import hip

    dims, lag, lb, steps, world = [64, 32, 16, 3], 5, 640, 4, 2
    X = ar_features(2 * 3000, dims[0], 21)
    Xn, _, _ = normalized(X)
    torch.manual_seed(4)
    acts = ["leaky_relu"] * (len(dims) - 2) + [None]
    seq = onn.feed_forward(dims, acts)
    lins = linears_of(seq)
    save = {f"w{i}": l.weight.detach().numpy() for i, l in enumerate(lins)}
    save.update({f"b{i}": l.bias.detach().numpy() for i, l in enumerate(lins)})
    np.savez(tmp_path / "dp.npz", Xn=Xn, dims=np.array(dims), lag=lag, lb=lb, steps=steps, **save)
    # single process: the same pairs per step (rank-major) through a gather index
    n_local = Xn.shape[0] // world
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=lb * world, lag=lag, tica_reg=1e-6, lr=1e-3)
    push_params(eng, lins)
    eng.reset_log(4 * steps + 8)
    Xd = torch.from_numpy(Xn).cuda()
    for i in range(steps):
        idx = torch.cat([torch.arange(r * n_local + i * lb, r * n_local + (i + 1) * lb) for r in range(world)]).cuda()
        eng.train_step(Xd, idx=idx)
    torch.cuda.synchronize()
    ref_lin = eng.get_linears()
    ref_loss = eng.read_log()[:steps, 0]
    eng.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f"dp_rank{r}.npz")
        np.testing.assert_allclose(z["loss"], ref_loss, rtol=1e-5, atol=1e-6)
        for i, (w, _) in enumerate(ref_lin):
            np.testing.assert_allclose(z[f"w{i}"], w, atol=2e-5)


def test_graphed_steps_match_plain_launches():
    """With step graphs switched on (dcv_mlp_set_graph / DCV_GRAPH=1), on a capturing-capable (non-null) stream a step is captured into a hipGraph whose
    instantiation is updated in place every call (dcv_mlp_train_step / forward / backward / eval_step); on the null
    stream the same kernels are launched one by one.  Same kernels, same order: the parameters after several steps, with changing batch offsets,
    an evaluation step in between and a changing batch size, must be bit-identical."""
    from deep_cartograph_amd import hip

    dims, lag = [64, 32, 16, 3], 5
    acts = ["leaky_relu", "leaky_relu", None]
    X = ar_features(6000, dims[0], 31)
    Xn, _, _ = normalized(X)
    torch.manual_seed(6)
    lins = linears_of(onn.feed_forward(dims, acts))
    Xd = torch.from_numpy(Xn).cuda()

    def run(stream):
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=1024, lag=lag, tica_reg=1e-6, lr=1e-3)
        push_params(eng, lins)
        eng.reset_log(64)
        eng.set_graph(True)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            for i in range(5):
                eng.train_step(Xd, row0=100 * i, batch=1024)
            eng.eval_step(Xd, row0=3000, batch=1024)
            for i in range(3):                       # split calls, as the data-parallel path makes them
                eng.forward(Xd, row0=50 * i, batch=768)
                eng.backward(Xd, row0=50 * i, batch=768, global_batch=768, train=True)
                eng.apply()
            eng.train_step(Xd, row0=7, batch=1024)
        torch.cuda.synchronize()
        out = eng.get_linears(), eng.read_log()[:10].copy(), eng.graph_launches()
        eng.close()
        return out

    plain, plain_log, n_plain = run(None)
    side = torch.cuda.Stream()
    graphed, graphed_log, n_graphed = run(side)
    assert n_plain == 0          # the default stream cannot capture
    assert n_graphed >= 8        # every call after the first of each kind went out as a graph
    for (w0, b0), (w1, b1) in zip(plain, graphed):
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(plain_log, graphed_log)


def test_native_rccl_communicator_world1():
    """dcv_comm_* (the library's own RCCL communicator) with one rank: creation from a unique id, an in-place all-reduce,
    and data-parallel steps through dcv_mlp_dp_step + dcv_comm_dp_allreduce_fn -- entirely inside the library -- against
    the same steps through the torch.distributed-free single-GPU path.  (No multi-GPU node is available to the builder:
    a multi-rank communicator has not executed; bench.py --native-rccl is the switch for a node that has one.)"""
    from deep_cartograph_amd import hip

    comm = hip.RcclComm(None)
    assert comm.world == 1 and comm.rank == 0
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.all_reduce(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    dims, lag, batch = [64, 32, 16, 3], 5, 1024
    acts = ["leaky_relu", "tanh", None]
    Xn, _, _ = normalized(ar_features(4000, dims[0], 3))
    torch.manual_seed(9)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    Xd = torch.from_numpy(Xn).cuda()
    res = {}
    for mode in ("native", "single", "native_overlap"):
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6, lr=1e-3)
        push_params(eng, linears_of(ref.nn))
        eng.reset_log(16)
        if mode == "native_overlap":
            os_env = __import__("os").environ
            os_env["DCV_DP_OVERLAP"] = "1"
        for i in range(5):
            if mode == "single":
                eng.train_step(Xd, row0=17 * i, batch=batch)
            else:
                eng.data_parallel_step(Xd, comm, batch, row0=17 * i, batch=batch, train=True)
        if mode != "single":
            eng.data_parallel_step(Xd, comm, batch, row0=2000, batch=batch, train=False)
        else:
            eng.eval_step(Xd, row0=2000, batch=batch)
        if mode == "native_overlap":
            os_env.pop("DCV_DP_OVERLAP", None)
        res[mode] = (eng.read_log()[:, 0].copy(), [w for w, _ in eng.get_linears()])
        eng.close()
    for mode in ("native", "native_overlap"):
        np.testing.assert_allclose(res[mode][0], res["single"][0], rtol=1e-5, atol=1e-6)
        for w, w0 in zip(res[mode][1], res["single"][1]):
            np.testing.assert_allclose(w, w0, atol=2e-6)
    comm.close()


_SWITCH_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from deep_cartograph_amd import hip
rng = np.random.Generator(np.random.PCG64(5))
model = sys.argv[1]
if model == "deep_tica":
    dims, acts = [96, 64, 32, 3], ["leaky_relu", "tanh", None]
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=2048, lag=7, tica_reg=1e-6, lr=1e-3)
else:
    dims, acts = [96, 48, 16, 2, 16, 48, 96], ["leaky_relu", "leaky_relu", None, "leaky_relu", "leaky_relu", None]
    eng = hip.Mlp("ae", dims, acts, max_batch=2048, latent_layer=3, lr=1e-3)
    eng.set_feature_range(np.ones(96, np.float32))
torch.manual_seed(3)
lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins])
X = torch.from_numpy(rng.standard_normal((12000, 96)).astype(np.float32).cumsum(0) * 0.02 + rng.standard_normal((12000, 96)).astype(np.float32)).cuda()
eng.reset_log(64)
for i in range(8):
    eng.train_step(X, row0=i * 1000, batch=2048 if i % 2 else 1999)
torch.cuda.synchronize()
h = hashlib.sha256()
for w, b in eng.get_linears():
    h.update(np.ascontiguousarray(w).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
h.update(np.ascontiguousarray(eng.read_log()[:8, 0]).tobytes())
print("HASH", h.hexdigest())
"""


@pytest.mark.parametrize("model", ["deep_tica", "ae"])
def test_launch_time_switches_leave_the_bits_alone(model, tmp_path):
    """The speed switches of the library change how results are stored or summed up in the SAME order, never the results:
    write-through epilogue stores (DCV_WT), the flat-grid gradient reduction (DCV_REDUCE_QUAD), the grouped weight /
    input gradient launch (DCV_NO_PAIR).  Eight training steps in fresh processes (the switches are read once per
    process), parameters and losses hashed."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "switch_run.py"
    script.write_text(_SWITCH_SCRIPT.format(root=root))
    hashes = {}
    for name, env in (("default", {}), ("plain stores", {"DCV_WT": "0"}), ("round-2 reduction", {"DCV_REDUCE_QUAD": "0"}),
                      ("two launches", {"DCV_NO_PAIR": "1"})):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, str(script), model], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        hashes[name] = [l for l in out.stdout.splitlines() if l.startswith("HASH")][-1]
    assert len(set(hashes.values())) == 1, hashes


def _fuzz_seeds():
    """The ten seeds of the suite, plus a range for one-off sweeps: DCV_FUZZ_SEEDS=100:200 (profiles/r04_gpu_calls.log has the
    sweeps that were run)."""
    extra = os.environ.get("DCV_FUZZ_SEEDS", "")
    if ":" in extra:
        lo, hi = extra.split(":")
        return list(range(10)) + list(range(int(lo), int(hi)))
    return list(range(10))


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_random_configurations_match_oracle(seed):
    """Random small configurations (widths 2..150 incl. odd ones, 1-3 hidden layers, batch 20..700, gathered or contiguous
    rows, both models, mixed activations): three SGD steps against the float32 autograd oracle.  Exercises whichever tile
    shapes, split plans, scalar / vector loaders and reduction items the shapes select."""
    from deep_cartograph_amd import hip

    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    model = "ae" if seed % 2 else "deep_tica"
    F = int(rng.integers(5, 150))
    hidden = [int(rng.integers(3, 120)) for _ in range(int(rng.integers(1, 4)))]
    d = int(rng.integers(1, 5))
    lag = int(rng.integers(1, 9))
    bs = int(rng.integers(20, 700))
    n = bs + lag + 64
    act_pool = ["leaky_relu", "tanh", "relu", "elu", "softplus"]
    X = ar_features(n, F, 300 + seed)
    Xn, m, r = normalized(X)
    torch.manual_seed(500 + seed)
    enc_dims = [F] + hidden + [d]
    enc_acts = [act_pool[int(rng.integers(0, len(act_pool)))] for _ in hidden] + [None]
    if model == "ae":
        dec_dims = enc_dims[::-1]
        dec_acts = [act_pool[int(rng.integers(0, len(act_pool)))] for _ in hidden] + [None]
        ref = onn.AEModel(enc_dims, enc_acts, None, dec_dims, dec_acts, None, m, r)
        dims, acts = enc_dims + dec_dims[1:], enc_acts + dec_acts
        eng = hip.Mlp("ae", dims, acts, max_batch=bs, latent_layer=len(enc_dims) - 1, optimizer="SGD", lr=5e-3, momentum=0.5)
        eng.set_feature_range(r)
        lins = linears_of(ref.encoder) + linears_of(ref.decoder)
        xt = torch.from_numpy(X)
    else:
        ref = onn.DeepTICAModel(enc_dims, enc_acts, None, None, None, 1e-6)
        dims, acts = enc_dims, enc_acts
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=bs, lag=lag, tica_reg=1e-6, optimizer="SGD", lr=5e-3, momentum=0.5)
        lins = linears_of(ref.nn)
        xt = torch.from_numpy(Xn)
    push_params(eng, lins)
    opt = torch.optim.SGD(ref.parameters(), lr=5e-3, momentum=0.5)
    Xd = torch.from_numpy(Xn).cuda()
    eng.reset_log(8)
    ref_losses = []
    for step in range(3):
        nb = bs if step != 1 else max(8, bs - int(rng.integers(1, 17)))
        start = int(rng.integers(0, 32))
        b = torch.arange(start, start + nb)
        opt.zero_grad()
        loss = ref.step(xt[b])[0] if model == "ae" else ref.step(xt[b], xt[b + lag])[0]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
        if (seed + step) % 2:
            eng.train_step(Xd, idx=b.cuda())
        else:
            eng.train_step(Xd, row0=start, batch=nb)
    log = eng.read_log()[:3, 0]
    assert np.all(np.isfinite(log)), (model, dims, acts, bs, log)
    scale = max(1.0, max(float(l.weight.abs().max()) for l in lins))
    dev_w = max(float(np.max(np.abs(w - lin.weight.detach().numpy()))) for (w, _), lin in zip(eng.get_linears(), lins)) / scale
    dev_l = float(np.max(np.abs(log - ref_losses) / np.maximum(np.abs(ref_losses), 1e-3)))
    print(f"random config {seed}: {model} {dims} {acts} batch {bs} lag {lag}: loss dev {dev_l:.1e}, weight dev {dev_w:.1e}")
    assert dev_l < 1e-5 and dev_w < 3e-7, (model, dims, acts, bs, lag, dev_l, dev_w)   # measured: <= 2.4e-6 / 6.0e-8
    eng.close()
