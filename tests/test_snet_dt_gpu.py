"""Fused small-network Deep-TICA kernels (snet_dt.hip): forward + batch statistics + loss head in one launch, backward in a
second, on the reference's own network sizes (cv_calculator.py:2569-2590; tools/train_colvars/default_config.yml:45-55).
Everything is checked against a FLOAT64 run of the autograd oracle on the same float32 parameters and inputs, through the
same C-ABI entry points as the layer-by-layer path (dcv_mlp_forward / _backward / _train_step / _eval_step)."""
import copy

import numpy as np
import pytest
import torch

from oracle import nn as onn
from tests.test_mlp_gpu import ar_features, linears_of, normalized, push_params, rel_err

pytestmark = pytest.mark.gpu


def _setup(dims, acts, n, lag, seed=3):
    X = ar_features(n, dims[0], 17)
    Xn, _, _ = normalized(X)
    torch.manual_seed(seed)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    return Xn, ref


@pytest.mark.parametrize("dims,hidden_act,last_act,n,lag,batch,gather", [
    ([54, 16, 8, 2], "leaky_relu", None, 900, 1, 128, True),       # the reference's test network, its clamped batch
    ([54, 15, 15, 2], "leaky_relu", None, 6000, 10, 4096, True),   # default_config.yml layers [15, 15]: widths padded to 16
    ([54, 15, 15, 2], "tanh", None, 6000, 10, 4096, False),        # contiguous batch: row sharing is given up, same numbers
    ([20, 7, 1], "relu", None, 400, 3, 37, False),                 # d = 1, a ragged last tile (37 = 2 * 16 + 5 pairs)
    ([128, 64, 32, 4], "tanh", None, 3000, 5, 1000, True),         # d = 4, wider layers, 63 tiles
    ([33, 12, 3], "elu", "tanh", 1500, 2, 515, True),              # scalar input loads (33 % 4 != 0), an activation on the outputs
    ([16, 2], None, None, 300, 1, 100, False),                     # a single Linear: no input gradient at all
])
def test_fused_step_matches_float64_autograd(dims, hidden_act, last_act, n, lag, batch, gather):
    from deep_cartograph_amd import hip

    acts = [hidden_act] * (len(dims) - 2) + [last_act]
    Xn, ref = _setup(dims, acts, n, lag)
    ref64 = copy.deepcopy(ref).double()
    P = Xn.shape[0] - lag
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
    push_params(eng, linears_of(ref.nn))
    Xd = torch.from_numpy(Xn).cuda()
    if gather:
        idx = torch.randperm(P)[:batch].contiguous()
        kw = dict(idx=idx.cuda())
    else:
        idx = torch.arange(5, 5 + batch)
        kw = dict(row0=5, batch=batch)
    eng.reset_log(4)
    eng.forward(Xd, **kw)
    assert eng.last_path() == 2, "the fused small-network kernels did not take this network"
    stats = eng.stats_view().cpu().numpy()
    eng.backward(Xd, **kw)
    g = eng.grads_view().cpu().numpy()
    eng.eval_step(Xd, **kw)          # evaluation step: forward + statistics + head, no blob, no gradient
    rec = eng.read_log()
    xt = torch.from_numpy(Xn).double()
    loss, _ = ref64.step(xt[idx], xt[idx + lag])
    loss.backward()
    with torch.no_grad():
        f_t = ref64.forward_nn(xt[idx])
        f_l = ref64.forward_nn(xt[idx + lag])
    d = dims[-1]
    np.testing.assert_allclose(stats[:d], f_t.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[d:2 * d], f_l.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[2 * d:2 * d + d * d].reshape(d, d), (f_t.T @ f_t).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(stats[2 * d + d * d:].reshape(d, d), (f_t.T @ f_l).numpy(), rtol=2e-5, atol=2e-5)
    assert len(rec) == 2 and rec[0, 1] == batch and rec[1, 1] == batch
    assert abs(rec[0, 0] - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert rec[1, 0] == rec[0, 0]   # the evaluation step of the same batch: the same partial sums in the same order
    lins = linears_of(ref64.nn)
    worst = 0.0
    for l, lin in enumerate(lins):
        wo, bo = eng.offsets[l]
        gw, gb = lin.weight.grad.numpy(), lin.bias.grad.numpy()
        ew = rel_err(g[wo:wo + gw.size].reshape(gw.shape), gw)
        worst = max(worst, ew)
        assert ew < 2e-5, f"layer {l} weight: {ew:.2e}"
        if l < len(lins) - 1 or last_act is not None:
            eb = rel_err(g[bo:bo + gb.size], gb)
            worst = max(worst, eb)
            assert eb < 2e-5 or np.max(np.abs(gb)) < 1e-9, f"layer {l} bias: {eb:.2e}"
        else:   # shift invariance of the loss: the exact gradient of the last bias is 0
            assert np.max(np.abs(g[bo:bo + gb.size])) < 2e-5 * max(1.0, np.max(np.abs(gw))), f"layer {l} bias"
    print(f"{dims} batch {batch}: worst gradient deviation from float64 = {worst:.2e}")
    eng.close()


def test_fused_training_follows_the_oracle():
    """40 Adam steps through dcv_mlp_train_step (fused forward + head, fused backward, reduction + Adam: three launches per
    step) against torch.optim.Adam over the float32 autograd oracle on the same shuffled batches; then a batch too large
    for the fused form (more than 512 tiles) falls back to the layer-by-layer path inside the same engine."""
    from deep_cartograph_amd import hip

    dims, acts, lag, batch = [54, 16, 8, 2], ["tanh", "tanh", None], 4, 256
    Xn, ref = _setup(dims, acts, 5000, lag, seed=9)
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=40000, lag=lag, tica_reg=1e-6, lr=2e-3)
    push_params(eng, linears_of(ref.nn))
    opt = torch.optim.Adam(ref.parameters(), lr=2e-3)
    Xd, Xt = torch.from_numpy(Xn).cuda(), torch.from_numpy(Xn)
    g = torch.Generator().manual_seed(5)
    eng.reset_log(64)
    for _ in range(40):
        idx = torch.randperm(Xn.shape[0] - lag, generator=g)[:batch].contiguous()
        eng.train_step(Xd, idx=idx.cuda())
        assert eng.last_path() == 2
        opt.zero_grad()
        loss, _ = ref.step(Xt[idx], Xt[idx + lag])
        loss.backward()
        opt.step()
    rec = eng.read_log()
    assert len(rec) == 40 and abs(rec[-1, 0] - float(loss)) < 3e-5 * max(1.0, abs(float(loss)))
    for (w, b), lin in zip(eng.get_linears(), linears_of(ref.nn)[:-1]):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=3e-6 * max(1.0, float(lin.weight.abs().max())))
        np.testing.assert_allclose(b, lin.bias.detach().numpy(), atol=3e-6)
    big = torch.from_numpy(normalized(ar_features(40100, 54, 3))[0]).cuda()
    eng.train_step(big, row0=0, batch=16384)    # 256 tiles of 128 rows: still the fused kernels
    assert eng.last_path() == 2
    eng.train_step(big, row0=0, batch=40000)    # 625 tiles: the general path
    assert eng.last_path() == 0
    eng.close()


def test_layer_output_hook_says_when_the_activations_never_left_lds():
    from deep_cartograph_amd import hip
    from deep_cartograph_amd._lib import DcvError

    dims, acts = [54, 16, 8, 2], ["tanh", "tanh", None]
    Xn, ref = _setup(dims, acts, 600, 2)
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=128, lag=2)
    push_params(eng, linears_of(ref.nn))
    eng.reset_log(2)
    eng.forward(torch.from_numpy(Xn).cuda(), row0=0, batch=128)
    with pytest.raises(DcvError, match="fused small-network"):
        eng.layer_output(0, 128)
    eng.close()


class _TwoRanksInOneProcess:
    """The slice of torch.distributed a data-parallel step touches, for TWO engines stepped one after the other in this
    process: rank A's call records its buffers, rank B's call adds them up and writes the sums into both -- possible
    because the collectives of dcv_mlp_dp_step run in a host callback between the library's launches.  (Real two-process
    runs: tests/test_mlp_gpu.py::test_data_parallel_two_ranks_match_single_process.)"""

    class ReduceOp:
        SUM = "sum"


@pytest.mark.parametrize("model", ["deep_tica", "ae"])
def test_fused_data_parallel_step_equals_the_single_process_step(model):
    """dcv_mlp_dp_step on the fused small-network kernels.  With ONE rank (global batch = local batch, all-reduces that
    change nothing) the data-parallel code path -- fused forward with the head left to the backward launch (Deep-TICA) or the
    fused step with the loss record written after the statistics exchange (autoencoder), reduction WITHOUT the fused
    optimiser, separate update -- must land on the weights of dcv_mlp_train_step: same partial sums, same update arithmetic."""
    from deep_cartograph_amd import hip

    class _OneRank(_TwoRanksInOneProcess):
        calls = []

        @staticmethod
        def all_reduce(t, op=None, group=None, async_op=False):
            _OneRank.calls.append(int(t.numel()))
            return None

    lag, batch = 3, 512
    if model == "deep_tica":
        dims, acts, kw = [54, 16, 8, 2], ["tanh", "tanh", None], dict(lag=lag)
    else:
        dims, acts, kw = [54, 16, 2, 16, 54], ["tanh", None, "tanh", None], dict(latent_layer=2)
    X = ar_features(3000, 54, 41)
    Xn, m, r = normalized(X)
    Xd = torch.from_numpy(Xn).cuda()
    torch.manual_seed(13)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    engs = []
    for _ in range(2):
        e = hip.Mlp(model, dims, acts, max_batch=batch, lr=2e-3, **kw)
        push_params(e, lins)
        if model == "ae":
            e.set_feature_range(r)
        e.reset_log(16)
        engs.append(e)
    g = torch.Generator().manual_seed(3)
    for step in range(5):
        idx = torch.randperm(Xn.shape[0] - lag, generator=g)[:batch].contiguous().cuda()
        engs[0].train_step(Xd, idx=idx)
        engs[1].data_parallel_step(Xd, _OneRank, batch, idx=idx, train=True)
        assert engs[0].last_path() == (2 if model == "deep_tica" else 1) and engs[1].last_path() == engs[0].last_path()
    engs[1].data_parallel_step(Xd, _OneRank, batch, idx=idx, train=False)   # evaluation step of the same batch
    engs[0].eval_step(Xd, idx=idx)
    ra, rb = engs[0].read_log(), engs[1].read_log()
    assert len(ra) == 6 and len(rb) == 6
    np.testing.assert_allclose(rb[:, 0], ra[:, 0], rtol=1e-12, atol=1e-12)   # same sums, same head arithmetic
    assert np.all(rb[:, 1] == batch)
    for (wa, ba), (wb, bb) in zip(engs[0].get_linears(), engs[1].get_linears()):
        np.testing.assert_array_equal(wb, wa)
        np.testing.assert_array_equal(bb, ba)
    assert len(_OneRank.calls) == 5 * 2 + 1     # statistics + gradients per training step, statistics for the evaluation step
    for e in engs:
        e.close()


@pytest.mark.parametrize("model,dims,batch,nb,gather", [
    ("deep_tica", [54, 16, 8, 2], 128, 5, True),      # the reference's network: 8 tiles per batch
    ("deep_tica", [54, 15, 15, 2], 1000, 7, False),   # a ragged last tile in every batch, consecutive rows
    ("deep_tica", [20, 7, 1], 37, 70, True),          # more batches than one launch takes (64): two launches
    ("ae", [54, 16, 8, 2], 128, 5, True),
    ("ae", [128, 64, 32, 2], 1000, 9, False),         # BASELINE C2's network
    ("ae", [33, 12, 3], 77, 66, True),
    ("deep_tica", [256, 512, 256, 3], 300, 3, True),  # too wide for the fused kernels: the entry point steps batch by batch
])
def test_batched_validation_pass_equals_step_by_step(model, dims, batch, nb, gather):
    """dcv_mlp_eval_steps: the records of nb evaluation steps from one call (small networks: many batches per launch),
    bit for bit those of nb dcv_mlp_eval_step calls, appended behind what the log already holds."""
    from deep_cartograph_amd import hip

    lag = 3
    n = batch * nb + 50
    X = ar_features(n + lag, dims[0], 5)
    Xn, _, _ = normalized(X)
    Xd = torch.from_numpy(Xn).cuda()
    torch.manual_seed(1)
    if model == "deep_tica":
        full = dims
        acts = ["tanh"] * (len(dims) - 2) + [None]
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
    else:
        full = dims + dims[-2::-1]
        acts = (["tanh"] * (len(dims) - 2) + [None]) * 2
        eng = hip.Mlp("ae", full, acts, max_batch=batch, latent_layer=len(dims) - 1)
        eng.set_feature_range(np.ones(dims[0], dtype=np.float32))
    push_params(eng, [torch.nn.Linear(full[i], full[i + 1]) for i in range(len(full) - 1)])
    idx = torch.randperm(n)[:batch * nb].contiguous().cuda() if gather else None
    # a log too short for the pass (the engine's first, 3 records): the records that fit are kept
    eng.reset_log(3)
    eng.eval_steps(Xd, batch, nb, idx=idx, row0=0 if gather else 7)
    c = eng.read_log()
    # step by step, behind one record so that the batched call starts at a non-zero counter
    def one_by_one():
        eng.reset_log(nb + 1)
        eng.eval_step(Xd, **(dict(idx=idx[:batch]) if gather else dict(row0=7, batch=batch)))
        for j in range(nb):
            kw = dict(idx=idx[j * batch:(j + 1) * batch]) if gather else dict(row0=7 + j * batch, batch=batch)
            eng.eval_step(Xd, **kw)
        return eng.read_log()
    a = one_by_one()
    eng.reset_log(nb + 1)
    eng.eval_step(Xd, **(dict(idx=idx[:batch]) if gather else dict(row0=7, batch=batch)))
    eng.eval_steps(Xd, batch, nb, idx=idx, row0=0 if gather else 7)
    b = eng.read_log()
    assert a.shape == b.shape == (nb + 1, eng.log_width)
    assert np.isfinite(a).all()
    assert len(np.unique(a[1:, 0])) == nb            # the batches do differ
    assert np.array_equal(a, b)
    assert np.array_equal(c, a[1:4])
    if dims[1] <= 64:
        assert eng.last_path() == (2 if model == "deep_tica" else 1)
    # and a training step still follows a batched pass
    eng.reset_log(2)
    eng.train_step(Xd, **(dict(idx=idx[:batch]) if gather else dict(row0=7, batch=batch)))
    assert np.isfinite(eng.read_log()).all()


@pytest.mark.parametrize("model,dims,batch,gather", [
    ("deep_tica", [54, 16, 8, 2], 128, True),
    ("ae", [54, 16, 8, 2], 100, False),
    ("deep_tica", [256, 512, 3], 300, True),   # the layer-by-layer engine
])
def test_training_steps_in_one_call_equal_step_by_step(model, dims, batch, gather):
    """dcv_mlp_train_steps: nsteps training steps behind one call -- parameters, optimiser state and loss records bit for bit
    those of nsteps dcv_mlp_train_step calls."""
    from deep_cartograph_amd import hip

    lag, nsteps = 2, 6
    n = batch * nsteps + 20
    Xn, _, _ = normalized(ar_features(n + lag, dims[0], 9))
    Xd = torch.from_numpy(Xn).cuda()
    idx = torch.randperm(n)[:batch * nsteps].contiguous().cuda() if gather else None

    def engine():
        torch.manual_seed(2)
        if model == "deep_tica":
            full, acts = dims, ["tanh"] * (len(dims) - 2) + [None]
            eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6, lr=1e-3)
        else:
            full, acts = dims + dims[-2::-1], (["tanh"] * (len(dims) - 2) + [None]) * 2
            eng = hip.Mlp("ae", full, acts, max_batch=batch, latent_layer=len(dims) - 1, lr=1e-3)
            eng.set_feature_range(np.ones(dims[0], dtype=np.float32))
        push_params(eng, [torch.nn.Linear(full[i], full[i + 1]) for i in range(len(full) - 1)])
        eng.reset_log(2 * nsteps)
        return eng

    a = engine()
    for j in range(nsteps):
        a.train_step(Xd, **(dict(idx=idx[j * batch:(j + 1) * batch]) if gather else dict(row0=3 + j * batch, batch=batch)))
    b = engine()
    b.train_steps(Xd, batch, nsteps, idx=idx, row0=0 if gather else 3)
    ra, rb = a.read_log(), b.read_log()
    assert ra.shape == (nsteps, a.log_width) and np.array_equal(ra, rb)
    assert torch.equal(a.params_view(), b.params_view())
    # and the next step continues from the same optimiser state
    kw = dict(idx=idx[:batch]) if gather else dict(row0=3, batch=batch)
    a.train_step(Xd, **kw)
    b.train_step(Xd, **kw)
    assert torch.equal(a.params_view(), b.params_view())


def test_epoch_entry_points_edge_cases():
    """dcv_mlp_train_steps / dcv_mlp_eval_steps at the edges: zero and one batch, an engine with dropout (no fused kernels: the
    validation pass steps batch by batch behind the same call, evaluation uses no masks), argument checks of the host wrapper."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd._lib import DcvError

    dims, batch, lag = [54, 16, 8, 2], 64, 2
    Xn, _, _ = normalized(ar_features(batch * 5 + 30, 54, 13))
    Xd = torch.from_numpy(Xn).cuda()
    for drops in (None, [0.25, 0.0, 0.0]):
        torch.manual_seed(3)
        eng = hip.Mlp("deep_tica", dims, ["tanh", "tanh", None], max_batch=batch, lag=lag, tica_reg=1e-6, dropout=drops, seed=5)
        push_params(eng, [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)])
        eng.reset_log(16)
        eng.train_steps(Xd, batch, 0, row0=0)
        eng.eval_steps(Xd, batch, 0, row0=0)
        assert len(eng.read_log()) == 0
        eng.eval_steps(Xd, batch, 1, row0=4)            # one batch: the single step
        eng.eval_step(Xd, row0=4, batch=batch)
        eng.eval_steps(Xd, batch, 3, row0=4)
        for j in range(3):
            eng.eval_step(Xd, row0=4 + j * batch, batch=batch)
        rec = eng.read_log()
        assert rec.shape[0] == 8 and np.array_equal(rec[0], rec[1]) and np.array_equal(rec[2:5], rec[5:8])
        assert eng.last_path() == (0 if drops else 2)
        eng.train_steps(Xd, batch, 2, row0=0)
        assert len(eng.read_log()) == 10 and np.isfinite(eng.read_log()).all()
        with pytest.raises(DcvError):
            eng.eval_steps(Xd, batch, 50, row0=0)       # rows beyond the matrix
        with pytest.raises(DcvError):
            eng.train_steps(Xd, batch, 3, idx=torch.arange(2 * batch).cuda())   # too few indices
