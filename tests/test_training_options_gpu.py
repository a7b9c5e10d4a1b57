"""Training options of the reference's YAML schema on the HIP engine: dropout, the
shifted_softplus / custom_sigmoid activations, torch.optim optimisers by name, learning-rate schedulers, a
separately supplied validation set, frame sharding of the tools.  Each against the CPU oracle.
Run on the GPU box: python -m pytest tests -m gpu"""
import copy
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import linear as ol
from oracle import nn as onn
from tests.test_calculators_gpu import TEST_COMMON, make_calc
from tests.test_mlp_gpu import ar_features, linears_of, normalized, push_params, rel_err

pytestmark = pytest.mark.gpu


def _engine_grads(eng, lins):
    g = eng.grads_view().cpu().numpy()
    out = []
    for l, lin in enumerate(lins):
        wo, bo = eng.offsets[l]
        out.append((g[wo:wo + lin.weight.numel()].reshape(tuple(lin.weight.shape)).copy(), g[bo:bo + lin.bias.numel()].copy()))
    return out


# ----------------------------------------------------------------------------- activations
@pytest.mark.parametrize("acts", [["shifted_softplus", "custom_sigmoid", None], ["custom_sigmoid", "tanh", "shifted_softplus"],
                                  ["elu", "softplus", None]])
def test_activation_step_matches_float64_autograd(acts):
    """mlcolvar's Shifted_Softplus / Custom_Sigmoid (yaml_schemas/train_colvars.py:25) through one Deep-TICA step."""
    from deep_cartograph_amd import hip

    dims, lag, batch = [48, 24, 12, 3], 4, 600
    Xn, _, _ = normalized(ar_features(2500, dims[0], 13))
    torch.manual_seed(8)
    ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6)
    ref64 = copy.deepcopy(ref).double()
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6)
    push_params(eng, linears_of(ref.nn))
    Xd = torch.from_numpy(Xn).cuda()
    eng.reset_log(2)
    eng.set_row_sharing(False)
    eng.forward(Xd, row0=3, batch=batch)
    eng.backward(Xd, row0=3, batch=batch)
    xt = torch.from_numpy(Xn).double()
    loss, _ = ref64.step(xt[3:3 + batch], xt[3 + lag:3 + lag + batch])
    loss.backward()
    assert abs(eng.read_log()[0, 0] - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    lins = linears_of(ref64.nn)
    for l, ((gw, gb), lin) in enumerate(zip(_engine_grads(eng, lins), lins)):
        assert rel_err(gw, lin.weight.grad.numpy()) < 5e-5, f"layer {l}"
    # inference path
    out, _ = eng.infer(Xd[:500])
    with torch.no_grad():
        exp = ref64.forward_nn(xt[:500]).numpy()
    np.testing.assert_allclose(out.cpu().numpy(), exp, atol=2e-5)
    eng.close()


# ----------------------------------------------------------------------------- dropout
def _queue_masks(model_seq, masks_per_call):
    """masks_per_call[c][k]: multiplier tensor of the k-th dropout module in forward call c."""
    drops = [m for m in model_seq if isinstance(m, onn.MaskedDropout)]
    for k, dmod in enumerate(drops):
        dmod.queue = [torch.from_numpy(call[k]).double() for call in masks_per_call]
    return drops


@pytest.mark.parametrize("acts,drops,gather", [
    (["leaky_relu", "leaky_relu", None], [0.1, 0.25, 0.0], False),   # sign-mask dgrad + fused head
    (["tanh", "relu", None], [0.2, 0.1, 0.0], True),                  # derivative from the stored (scaled) output
    (["leaky_relu", "elu", None], [0.1, 0.1, 0.3], False),            # dropout on the network output: general kernels
])
def test_deeptica_dropout_step_matches_oracle_given_the_masks(acts, drops, gather):
    """torch.nn.Dropout in a training step (reference default_config.yml: dropout [0.1, 0.1]).  The engine draws its
    masks from a counter-based generator; given those very masks (dcv_mlp_dropout_mask) the float64 oracle must
    produce the same loss and gradients, the keep rate must match p, evaluation mode must not drop anything."""
    from deep_cartograph_amd import hip

    dims, lag, batch = [64, 128, 32, 3], 5, 900
    Xn, _, _ = normalized(ar_features(3000, dims[0], 17))
    torch.manual_seed(12)
    ref = onn.DeepTICAModel(dims, acts, [d if d else None for d in drops], None, None, 1e-6)
    ref64 = copy.deepcopy(ref).double()
    eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6, dropout=drops, seed=77)
    push_params(eng, linears_of(ref.nn))
    Xd = torch.from_numpy(Xn).cuda()
    if gather:
        idx = torch.randperm(Xn.shape[0] - lag)[:batch].contiguous()
        kw = dict(idx=idx.cuda())
    else:
        idx = torch.arange(11, 11 + batch)
        kw = dict(row0=11, batch=batch)
    eng.reset_log(4)
    eng.forward(Xd, train=True, **kw)     # training step 0
    eng.backward(Xd, **kw)
    assert eng.dropout_step() == 1
    # with dropout the two halves are separate rows: [0, B) x_t, [B, 2B) x_lag
    layers = [l for l, p in enumerate(drops) if p > 0]
    masks = {l: eng.dropout_mask(l, 0, 2 * batch).cpu().numpy() for l in layers}
    for l in layers:
        keep = np.mean(masks[l] != 0)
        assert abs(keep - (1 - drops[l])) < 4 * np.sqrt(drops[l] * (1 - drops[l]) / masks[l].size) + 1e-3, (l, keep)
        assert set(np.unique(masks[l]).tolist()) <= {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(drops[l])))}
    ref64.train()
    _queue_masks(ref64.nn, [[masks[l][:batch] for l in layers], [masks[l][batch:] for l in layers]])
    xt = torch.from_numpy(Xn).double()
    loss, _ = ref64.step(xt[idx], xt[idx + lag])
    loss.backward()
    rec = eng.read_log()
    assert abs(rec[0, 0] - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    lins = linears_of(ref64.nn)
    for l, ((gw, gb), lin) in enumerate(zip(_engine_grads(eng, lins), lins)):
        assert rel_err(gw, lin.weight.grad.numpy()) < 5e-5, f"layer {l} weight"
        if l < len(lins) - 1:
            assert rel_err(gb, lin.bias.grad.numpy()) < 5e-5, f"layer {l} bias"
    # a different step draws a different mask; evaluation mode equals the dropout-free network
    m1 = eng.dropout_mask(layers[0], 1, 64).cpu().numpy()
    assert np.mean(m1 != masks[layers[0]][:64]) > 0.05
    eng.eval_step(Xd, **kw)
    ref64.eval()
    with torch.no_grad():
        ev_loss, _ = ref64.step(xt[idx], xt[idx + lag])
    assert abs(eng.read_log()[1, 0] - float(ev_loss)) < 2e-5 * max(1.0, abs(float(ev_loss)))
    eng.close()


def test_ae_dropout_step_matches_oracle_given_the_masks():
    from deep_cartograph_amd import hip

    F, batch = 40, 700
    enc, dec = [F, 24, 8, 2], [2, 8, 24, F]
    acts_e, acts_d = ["leaky_relu", "tanh", None], ["leaky_relu", "leaky_relu", None]
    drops = [0.1, 0.2, 0.0, 0.15, 0.0, 0.0]
    X = ar_features(2000, F, 3)
    Xn, m, r = normalized(X)
    torch.manual_seed(2)
    none_if0 = lambda ps: [p if p else None for p in ps]
    ref = onn.AEModel(enc, acts_e, none_if0(drops[:3]), dec, acts_d, none_if0(drops[3:]), m, r)
    ref64 = copy.deepcopy(ref).double()
    eng = hip.Mlp("ae", enc + dec[1:], acts_e + acts_d, max_batch=batch, latent_layer=3, dropout=drops, seed=5)
    push_params(eng, linears_of(ref.encoder) + linears_of(ref.decoder))
    eng.set_feature_range(r)
    Xd = torch.from_numpy(Xn).cuda()
    eng.reset_log(2)
    eng.forward(Xd, row0=100, batch=batch, train=True)
    eng.backward(Xd, row0=100, batch=batch)
    layers = [l for l, p in enumerate(drops) if p > 0]
    masks = {l: eng.dropout_mask(l, 0, batch).cpu().numpy() for l in layers}
    ref64.train()
    _queue_masks(ref64.encoder, [[masks[l] for l in layers if l < 3]])
    _queue_masks(ref64.decoder, [[masks[l] for l in layers if l >= 3]])
    loss, _ = ref64.step(torch.from_numpy(X[100:100 + batch]).double())
    loss.backward()
    assert abs(eng.read_log()[0, 0] - float(loss)) < 5e-5 * max(1.0, abs(float(loss)))
    lins = linears_of(ref64.encoder) + linears_of(ref64.decoder)
    for l, ((gw, gb), lin) in enumerate(zip(_engine_grads(eng, lins), lins)):
        assert rel_err(gw, lin.weight.grad.numpy()) < 1e-4, f"layer {l} weight"
        assert rel_err(gb, lin.bias.grad.numpy()) < 1e-4, f"layer {l} bias"
    eng.close()


# ----------------------------------------------------------------------------- batch normalisation
def _bn_modules(seq):
    return [m for m in seq if isinstance(m, torch.nn.BatchNorm1d)]


@pytest.mark.parametrize("kind", ["ae", "deep_tica"])
def test_batchnorm_training_matches_torch(kind):
    """`batchnorm` / `last_layer_batchnorm` of the YAML (cv_calculator.py:1155-1219, yaml_schemas/train_colvars.py:24-31):
    torch.nn.BatchNorm1d behind a Linear (after activation and dropout).  Training mode: batch statistics -- for Deep-TICA of
    the x_t call and of the x_lag call separately, x_t first -- and running-statistics updates; evaluation / inference: the
    running statistics.  (i) the first step's gradients against a FLOAT64 run of the autograd oracle; (ii) eight SGD-with-momentum steps
    against the float32 oracle: losses, every parameter incl. the normalisations' weight / bias, running mean / variance,
    num_batches_tracked; (iii) an evaluation step and the inference path in eval mode."""
    from deep_cartograph_amd import hip

    F, lag, batch, d = 40, 3, 384, 2
    Xn, m, r = normalized(ar_features(2600, F, 17))
    torch.manual_seed(12)
    if kind == "ae":
        enc, dec = [F, 24, d], [d, 24, F]
        acts = ["tanh", None, "leaky_relu", None]
        bnf = [True, False, True, True]            # incl. the decoder's output layer (last_layer_batchnorm)
        ref = onn.AEModel(enc, acts[:2], None, dec, acts[2:], None, None, None, enc_bn=bnf[:2], dec_bn=bnf[2:])
        dims, seqs, latent = enc + dec[1:], [ref.encoder, ref.decoder], 2
        eng = hip.Mlp("ae", dims, acts, max_batch=batch, latent_layer=latent, optimizer="SGD", lr=0.05, momentum=0.5, batchnorm=bnf)
        eng.set_feature_range(np.ones(F, np.float32))
    else:
        # (no normalisation behind the LAST Deep-TICA layer here: the loss is invariant under any invertible linear map of the
        # outputs, so that layer's weight and bias would have exactly zero gradients and Adam would move them by rounding noise)
        dims, acts, bnf = [F, 24, 12, d], ["leaky_relu", "tanh", None], [True, True, False]
        ref = onn.DeepTICAModel(dims, acts, None, None, None, 1e-6, batchnorm=bnf)
        seqs = [ref.nn]
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=batch, lag=lag, tica_reg=1e-6, optimizer="SGD", lr=0.05, momentum=0.5, batchnorm=bnf)
    lins = [m_ for q in seqs for m_ in linears_of(q)]
    bns = [m_ for q in seqs for m_ in _bn_modules(q)]
    with torch.no_grad():   # not the trivial weight 1 / bias 0
        for b in bns:
            b.weight.uniform_(0.5, 1.5)
            b.bias.uniform_(-0.3, 0.3)
    bn_state = []
    it = iter(bns)
    for flag in bnf:
        if flag:
            b = next(it)
            bn_state.append({"weight": b.weight.detach().numpy(), "bias": b.bias.detach().numpy(), "running_mean": b.running_mean.numpy(),
                             "running_var": b.running_var.numpy(), "num_batches_tracked": 0})
        else:
            bn_state.append(None)
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins], bn=bn_state)
    Xd = torch.from_numpy(Xn).cuda()
    xt = torch.from_numpy(Xn)

    def ref_step(model, x, r0):
        if kind == "ae":
            return model.step(x[r0:r0 + batch])[0]
        return model.step(x[r0:r0 + batch], x[r0 + lag:r0 + lag + batch])[0]

    # (i) first step, float64
    ref64 = copy.deepcopy(ref).double().train()
    eng.reset_log(32)
    eng.forward(Xd, row0=5, batch=batch, train=True)
    eng.backward(Xd, row0=5, batch=batch)
    loss64 = ref_step(ref64, xt.double(), 5)
    loss64.backward()
    assert abs(eng.read_log()[0, 0] - float(loss64)) < 2e-5 * max(1.0, abs(float(loss64)))
    g = eng.grads_view().cpu().numpy()
    lins64 = [m_ for q in ([ref64.encoder, ref64.decoder] if kind == "ae" else [ref64.nn]) for m_ in linears_of(q)]
    bns64 = [m_ for q in ([ref64.encoder, ref64.decoder] if kind == "ae" else [ref64.nn]) for m_ in _bn_modules(q)]
    worst = 0.0
    gscale = max(float(np.max(np.abs(lin.weight.grad.numpy()))) for lin in lins64)

    def cmp(got, exp):
        """relative deviation -- or, where the float64 gradient is exactly zero by an invariance of the loss (a constant shift
        in front of a normalisation or of Deep-TICA's mean removal: its bias, or the bias of a normalisation feeding such a
        Linear), the size of the engine's rounding noise against the largest gradient of the step"""
        nonlocal worst
        if np.max(np.abs(exp)) < 1e-12:
            assert np.max(np.abs(got)) < 1e-6 * gscale, (np.max(np.abs(got)), gscale)
            return
        worst = max(worst, rel_err(got, exp))

    for l, lin in enumerate(lins64):
        wo, bo = eng.offsets[l]
        gw, gb = lin.weight.grad.numpy(), lin.bias.grad.numpy()
        cmp(g[wo:wo + gw.size].reshape(gw.shape), gw)
        cmp(g[bo:bo + gb.size], gb)
    it64 = iter(bns64)
    for l, flag in enumerate(bnf):
        if flag:
            b = next(it64)
            go, bo_ = eng.bn_offsets[l]
            cmp(g[go:go + b.weight.numel()], b.weight.grad.numpy())
            cmp(g[bo_:bo_ + b.bias.numel()], b.bias.grad.numpy())
    print(f"{kind} batchnorm step: worst gradient deviation from float64 = {worst:.2e}")
    assert worst < 5e-5
    # (ii) training steps, float32 oracle (fresh engine state: the forward above already updated the running statistics)
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in lins], bn=bn_state)
    # SGD with momentum: the update is linear in the gradient.  (Adam divides by |g|: parameters whose exact gradient is zero
    # by an invariance -- see cmp above -- take +-lr steps of rounding noise in ANY float32 implementation)
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.5)
    ref.train()
    eng.reset_log(32)
    ref_losses = []
    for i in range(8):
        r0 = (i * 131) % (Xn.shape[0] - batch - lag)
        eng.train_step(Xd, row0=r0, batch=batch)
        opt.zero_grad()
        loss = ref_step(ref, xt, r0)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    np.testing.assert_allclose(eng.read_log()[:, 0], ref_losses, rtol=2e-4, atol=2e-5)
    for (w, b), lin in zip(eng.get_linears(), lins):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=5e-5)
    got_bn = [b for b in eng.get_bn() if b is not None]
    for gb_, b in zip(got_bn, bns):
        np.testing.assert_allclose(gb_["weight"], b.weight.detach().numpy(), atol=5e-5)
        np.testing.assert_allclose(gb_["running_mean"], b.running_mean.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gb_["running_var"], b.running_var.numpy(), rtol=1e-4, atol=1e-6)
        assert gb_["num_batches_tracked"] == int(b.num_batches_tracked) == (8 if kind == "ae" else 16)
    # (iii) evaluation mode
    ref.eval()
    eng.reset_log(4)
    eng.eval_step(Xd, row0=700, batch=batch)
    with torch.no_grad():
        le = float(ref_step(ref, xt, 700))
        exp = (ref.forward_cv(xt[:600]) if kind == "ae" else ref.forward_nn(xt[:600])).numpy()
    assert abs(eng.read_log()[0, 0] - le) < 2e-4 * max(1.0, abs(le))
    out, _ = eng.infer(Xd[:600])
    np.testing.assert_allclose(out.cpu().numpy(), exp, atol=5e-5)
    assert [b["num_batches_tracked"] for b in eng.get_bn() if b is not None] == [int(b.num_batches_tracked) for b in bns]   # eval steps track nothing
    eng.close()


def test_batchnorm_calculator_and_export(features, tmp_path):
    """A Deep-TICA calculator with batchnorm [True, False] + last_layer_batchnorm: the fit runs, the exported TorchScript
    carries the BatchNorm1d modules (names of the reference's module tree) and, loaded with plain torch.jit, reproduces
    the calculator's projection; a model.zip round trip through CVCalculator.load projects identically."""
    import io
    import zipfile

    from deep_cartograph_amd.cv_calculator import CVCalculator

    X, names = features
    arch = json.loads(json.dumps(TEST_COMMON["architecture"]))
    arch["encoder"]["batchnorm"] = [True, False]
    arch["encoder"]["last_layer_batchnorm"] = True
    calc = make_calc("deep_tica", tmp_path, architecture=arch)
    calc.set_training_matrix(X.copy(), names)
    df = calc.run(2)
    assert df is not None and df.shape == (164, 2)
    with zipfile.ZipFile(tmp_path / "deep_tica" / "model.zip") as z:
        ts = torch.jit.load(io.BytesIO(z.read("model/cv_weights.pt")))
    bnames = sorted(n for n, _ in ts.named_buffers())
    assert sum(n.endswith("running_mean") for n in bnames) == 2 and sum(n.endswith("num_batches_tracked") for n in bnames) == 2, bnames
    assert all(n.startswith("nn.nn.") for n in bnames if "running" in n)   # inside the FeedForward's Sequential, as the reference's tree
    with torch.no_grad():
        np.testing.assert_allclose(ts(torch.from_numpy(X)).numpy(), df.to_numpy(), atol=5e-5)
    loaded = CVCalculator.load(str(tmp_path / "deep_tica" / "model.zip"), str(tmp_path / "reload"))
    np.testing.assert_allclose(loaded.project_data(torch.from_numpy(X.copy())).numpy(), df.to_numpy(), atol=5e-5)


# ----------------------------------------------------------------------------- optimisers
@pytest.mark.parametrize("name,kwargs", [
    ("Adam", dict(lr=2e-3, weight_decay=1e-3)),
    ("Adam", dict(lr=1e-3, amsgrad=True, betas=(0.8, 0.99))),
    ("AdamW", dict(lr=2e-3, weight_decay=0.05)),
    ("SGD", dict(lr=0.05)),
    ("SGD", dict(lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-4)),
    ("SGD", dict(lr=0.02, momentum=0.8, dampening=0.1)),
    ("RMSprop", dict(lr=1e-3, momentum=0.5, centered=True)),
    ("RMSprop", dict(lr=2e-3, alpha=0.9, weight_decay=1e-3)),
    ("Adagrad", dict(lr=0.02, lr_decay=0.01, initial_accumulator_value=0.1)),
    ("Adamax", dict(lr=2e-3, weight_decay=1e-3)),
    ("NAdam", dict(lr=2e-3, momentum_decay=4e-3)),
    ("NAdam", dict(lr=1e-3, weight_decay=0.01, decoupled_weight_decay=True)),
    ("RAdam", dict(lr=2e-3)),                       # 12 steps: rho_t crosses 5 at step 6 -- both branches run
    ("RAdam", dict(lr=1e-3, weight_decay=0.01, decoupled_weight_decay=True, betas=(0.9, 0.9))),
    ("Adadelta", dict(lr=1.0, rho=0.9, weight_decay=1e-4)),
    ("ASGD", dict(lr=0.02, lambd=1e-3, alpha=0.75)),
    ("Rprop", dict(lr=1e-3, etas=(0.5, 1.2), step_sizes=(1e-6, 1e-2))),
    ("Adam", dict(lr=1e-3, maximize=True)),             # torch.optim's maximize: the negated gradient drives the update
    ("SGD", dict(lr=1e-4, weight_decay=1e-3, maximize=True)),      # (a gentle ascent: with momentum the maximised loss diverges within the 12 steps)
])
def test_optimizers_follow_torch(name, kwargs):
    """optimizer.name / kwargs of the YAML (cv_calculator.py:1377-1380 -> getattr(torch.optim, name)): 12 AE steps on the
    engine against the same steps of torch's optimiser over the autograd oracle (float32 both; 2e-5 on the weights)."""
    from deep_cartograph_amd import hip

    F, batch = 24, 512
    enc, dec = [F, 12, 2], [2, 12, F]
    acts = ["tanh", None, "tanh", None]
    X = ar_features(1600, F, 23)
    Xn, m, r = normalized(X)
    torch.manual_seed(31)
    ref = onn.AEModel(enc, acts[:2], None, dec, acts[2:], None, m, r)
    lins = linears_of(ref.encoder) + linears_of(ref.decoder)
    from deep_cartograph_amd.cv_calculator import NonLinear, _OPTIMIZERS

    ek = NonLinear._engine_optimizer_kwargs(name, {**_OPTIMIZERS[name], **kwargs})   # the calculators' own mapping onto the engine
    ek.pop("optimizer")
    eng = hip.Mlp("ae", enc + dec[1:], acts, max_batch=batch, latent_layer=2, optimizer=name, **ek)
    push_params(eng, lins)
    eng.set_feature_range(r)
    opt = getattr(torch.optim, name)(ref.parameters(), **kwargs)
    Xd = torch.from_numpy(Xn).cuda()
    Xt = torch.from_numpy(X)
    eng.reset_log(16)
    for i in range(12):
        r0 = (i * 97) % (X.shape[0] - batch)
        eng.train_step(Xd, row0=r0, batch=batch)
        opt.zero_grad()
        loss, _ = ref.step(Xt[r0:r0 + batch])
        loss.backward()
        opt.step()
    got = eng.get_linears()
    losses = eng.read_log()[:, 0]
    assert abs(losses[-1] - float(loss)) < 2e-4 * abs(float(loss))
    for (w, b), lin in zip(got, lins):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=2e-5, rtol=2e-4)
        np.testing.assert_allclose(b, lin.bias.detach().numpy(), atol=2e-5, rtol=2e-4)
    eng.close()


@pytest.mark.parametrize("name,kwargs", [
    ("Rprop", dict(lr=1e-3, etas=(0.5, 1.2), step_sizes=(1e-6, 1e-2))),
    ("ASGD", dict(lr=0.02, lambd=1e-3, alpha=0.75)),
])
def test_lazy_optimizer_state_sees_the_schedulers_lr(name, kwargs):
    """torch creates Rprop's step_size and ASGD's eta inside the FIRST optimizer.step(), from the learning rate
    the group holds then -- after a scheduler's constructor has rescaled it (OneCycleLR starts at max_lr / 25).  The
    engine seeds them right before its first update from the lr set through dcv_mlp_set_lr, not from the lr of creation."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd.cv_calculator import NonLinear, _OPTIMIZERS

    F, batch, steps = 24, 512, 10
    enc, dec = [F, 12, 2], [2, 12, F]
    acts = ["tanh", None, "tanh", None]
    X = ar_features(1600, F, 29)
    Xn, m, r = normalized(X)
    torch.manual_seed(37)
    ref = onn.AEModel(enc, acts[:2], None, dec, acts[2:], None, m, r)
    lins = linears_of(ref.encoder) + linears_of(ref.decoder)
    ek = NonLinear._engine_optimizer_kwargs(name, {**_OPTIMIZERS[name], **kwargs})
    ek.pop("optimizer")
    eng = hip.Mlp("ae", enc + dec[1:], acts, max_batch=batch, latent_layer=2, optimizer=name, **ek)
    push_params(eng, lins)
    eng.set_feature_range(r)
    opt = getattr(torch.optim, name)(ref.parameters(), **kwargs)
    # (ASGD with lr up to 0.4 diverges on this problem -- weights of 1e4 after ten steps, where two float32 runs part ways)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=(20 if name == "Rprop" else 2) * kwargs["lr"], total_steps=steps, cycle_momentum=False)
    assert abs(opt.param_groups[0]["lr"] - kwargs["lr"]) > 0.1 * kwargs["lr"]   # the constructor has already moved the lr
    Xd = torch.from_numpy(Xn).cuda()
    Xt = torch.from_numpy(X)
    eng.reset_log(16)
    for i in range(steps):
        r0 = (i * 97) % (X.shape[0] - batch)
        eng.set_lr(opt.param_groups[0]["lr"])
        eng.train_step(Xd, row0=r0, batch=batch)
        opt.zero_grad()
        loss, _ = ref.step(Xt[r0:r0 + batch])
        loss.backward()
        opt.step()
        sched.step()
    for (w, b), lin in zip(eng.get_linears(), lins):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=2e-5, rtol=2e-4)
        np.testing.assert_allclose(b, lin.bias.detach().numpy(), atol=2e-5, rtol=2e-4)
    eng.close()


def test_batchnorm_is_refused_in_a_data_parallel_step():
    """a frame-sharded step with batch normalisation would normalise with each rank's local rows (and let the
    running statistics of the ranks drift apart): dcv_mlp_dp_step refuses it with a message instead of computing something
    that is no longer the single-process fit.  One rank (global batch = local batch) is the single-process fit and runs."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd._lib import DcvError

    class _OneRank:   # the slice of torch.distributed the step touches
        class ReduceOp:
            SUM = "sum"

        @staticmethod
        def all_reduce(t, op=None, group=None, async_op=False):
            return None

    F, batch = 16, 256
    X = torch.from_numpy(ar_features(1200, F, 5)).cuda()
    eng = hip.Mlp("deep_tica", [F, 8, 2], ["tanh", None], max_batch=batch, lag=3, batchnorm=[True, False])
    push_params(eng, linears_of(torch.nn.Sequential(torch.nn.Linear(F, 8), torch.nn.Linear(8, 2))))
    eng.reset_log(8)
    with pytest.raises(DcvError, match="batch normalisation is not implemented for data-parallel"):
        eng.data_parallel_step(X, _OneRank, 2 * batch, row0=0, batch=batch, train=True)
    eng.data_parallel_step(X, _OneRank, batch, row0=0, batch=batch, train=True)   # one rank: allowed
    assert np.isfinite(eng.read_log()[:, 0]).all()
    eng.close()


# ----------------------------------------------------------------------------- calculators: schedulers, validation set
def _training(**general):
    t = json.loads(json.dumps(TEST_COMMON["training"]))
    t["general"].update(general)
    return t


def _oracle_ae(X, m, r, **kw):
    base = dict(seed_try=43, lengths=[0.8, 0.2], batch_size=32, shuffle=False, random_split=True, max_epochs=30, check_val_every_n_epoch=1,
                save_check_every_n_epoch=1, patience=50, min_delta=1e-5, opt_kwargs={"lr": 1e-3, "weight_decay": 0}, model_to_save="last")
    base.update(kw)
    return onn.train(None, {"data": torch.from_numpy(X)}, build_model=lambda: onn.AEModel(
        [54, 16, 8, 2], ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None], [2, 4, 8, 54], ["leaky_relu", "leaky_relu", None],
        [0.0, 0.0, None], m, r), **base)


@pytest.mark.parametrize("sched,config", [
    ({"name": "OneCycleLR", "kwargs": {"max_lr": 5e-3}}, {"interval": "epoch", "monitor": "valid_loss", "frequency": 1}),
    ({"name": "ReduceLROnPlateau", "kwargs": {"factor": 0.5, "patience": 1, "cooldown": 0, "threshold": 0.2}},
     {"interval": "epoch", "monitor": "valid_loss", "frequency": 1}),
    ({"name": "StepLR", "kwargs": {"step_size": 5, "gamma": 0.5}}, None),
    ({"name": "CosineAnnealingLR", "kwargs": {"T_max": 30}}, {"interval": "epoch", "frequency": 2}),
])
def test_lr_schedulers_follow_the_oracle(features, tmp_path, sched, config):
    """lr_scheduler of the YAML (reference :1228-1273, :1382-1394): the torch scheduler class runs on the host and drives
    the engine's learning rate (and beta1 for OneCycleLR); weights after 30 epochs against the oracle with the same
    scheduler attached to its torch optimiser (stepping per lightning's interval / frequency)."""
    X, names = features
    tr = _training(batch_size=32, max_epochs=30)
    tr["early_stopping"]["patience"] = 50
    tr["lr_scheduler"] = sched
    tr["lr_scheduler_config"] = config
    calc = make_calc("ae", tmp_path, training=tr)
    calc.set_training_matrix(X.copy(), names)
    assert calc.train()
    m, r = calc.features_norm_mean.astype(np.float32), calc.features_norm_range.astype(np.float32)
    n_train = 132   # 164 frames: 132 / 32
    so = calc._scheduler_options((n_train + 31) // 32)
    res = _oracle_ae(X, m, r, scheduler={"name": so[0], "kwargs": so[1], "config": so[2]})
    assert len(calc.metrics["epoch"]) == len(res["metrics"]["epoch"]) == 30
    np.testing.assert_allclose(calc.metrics["valid_loss"], res["metrics"]["valid_loss"], rtol=5e-4)
    assert "lr" in calc.metrics and len(calc.metrics["lr"]) == 30
    lins = linears_of(res["model"].encoder) + linears_of(res["model"].decoder)
    for (w, b), lin in zip(calc.cv["linears"], lins):
        np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=2e-4)


def test_separate_validation_set(features, tmp_path):
    """val_colvars_paths (reference :1485-1492, :2546-2553): train on ALL training samples, evaluate on the validation
    frames normalised with the training statistics; Deep-TICA pairs the validation frames among themselves.
    Deep-TICA runs with tanh layers here: with (leaky-)ReLU layers and small sequential batches some hidden units stay on
    one side of their kink over a whole batch, their bias gradient is then EXACTLY zero (the loss gradient rows sum to
    zero), float32 leaves 1e-7 of rounding noise there in torch as in the engine, and Adam turns noise above its 1e-8
    eps into full +-lr steps of random sign -- two float32 runs of the reference itself part ways at once on such a
    configuration ("noise-driven parameters")."""
    X, names = features
    Xtr, Xva = X[:120].copy(), X[120:].copy()
    for cv, act in (("ae", "leaky_relu"), ("deep_tica", "tanh")):
        tr = _training(batch_size=32, max_epochs=15)
        arch = json.loads(json.dumps(TEST_COMMON["architecture"]))
        arch["encoder"]["activation"] = [act, act]
        calc = make_calc(cv, tmp_path / cv, training=tr, architecture=arch)
        calc.set_training_matrix(Xtr.copy(), names)
        calc.set_validation_matrix(Xva.copy())
        assert calc.train()
        assert (calc.num_training_samples, calc.num_validation_samples) == (120, 44)
        m, r = calc.features_norm_mean.astype(np.float32), calc.features_norm_range.astype(np.float32)
        kw = dict(seed_try=43, batch_size=32, shuffle=False, max_epochs=15, patience=20, opt_kwargs={"lr": 1e-3, "weight_decay": 0},
                  model_to_save="last")
        tt, tv = torch.from_numpy(Xtr), torch.from_numpy(Xva)
        if cv == "ae":
            res = onn.train(None, {"data": tt}, val_data={"data": tv}, build_model=lambda: onn.AEModel(
                [54, 16, 8, 2], [act, act, None], [0.0, 0.0, None], [2, 4, 8, 54], ["leaky_relu", "leaky_relu", None],
                [0.0, 0.0, None], m, r), **kw)
            lins = linears_of(res["model"].encoder) + linears_of(res["model"].decoder)
        else:
            res = onn.train(None, {"data": tt[:-1], "data_lag": tt[1:]}, val_data={"data": tv[:-1], "data_lag": tv[1:]},
                            build_model=lambda: onn.DeepTICAModel([54, 16, 8, 2], [act, act, None], [0.0, 0.0, None], m, r, 1e-6), **kw)
            lins = linears_of(res["model"].nn)
        np.testing.assert_allclose(calc.metrics["valid_loss"], res["metrics"]["valid_loss"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(calc.metrics["train_loss"], res["metrics"]["train_loss"], rtol=1e-3, atol=1e-4)
        for (w, b), lin in zip(calc.cv["linears"], lins):
            np.testing.assert_allclose(w, lin.weight.detach().numpy(), atol=1e-4)


def test_unsupported_options_skip_the_cv_like_a_failed_fit(features, tmp_path):
    """The two torch.optim classes the engine does not run (LBFGS: closure-driven; SparseAdam: rejects dense gradients inside
    torch itself) are refused explicitly: the try is logged as failed and run() returns None -- what the reference does
    with any exception inside a try (cv_calculator.py:1541-1542, :407-409).  Every other optimiser name and batch
    normalisation run (test_optimizers_follow_torch, test_batchnorm_*)."""
    X, names = features
    for name in ("LBFGS", "SparseAdam"):
        tr = _training(max_epochs=2)
        tr["optimizer"] = {"name": name, "kwargs": {"lr": 1e-3}}
        calc = make_calc("ae", tmp_path / name, training=tr)
        calc.set_training_matrix(X.copy(), names)
        assert calc.run(2) is None
    tr = _training(max_epochs=2)
    tr["optimizer"] = {"name": "NoSuchOptimizer", "kwargs": {}}
    calc = make_calc("ae", tmp_path / "unknown", training=tr)
    calc.set_training_matrix(X.copy(), names)
    assert calc.run(2) is None


def test_reference_default_yaml_values_run(features, tmp_path):
    """The values of the reference's shipped tools/train_colvars/default_config.yml (dimension 1, lag 10, encoder layers
    [15, 15] with dropout [0.1, 0.1] and the schema's 3-entry activation default, batch 32, check / save every 10 epochs,
    patience 2, Adam lr 0.01; AE: shuffle + random split) through tools.train_colvars on the GPU, max_epochs cut to 60."""
    from deep_cartograph_amd import colvars, tools

    X, names = features
    path = str(tmp_path / "train.dat")
    colvars.write_colvars(path, X, names)
    cfg = {
        "cvs": ["pca", "ae", "tica", "deep_tica"],
        "common": {
            "dimension": 1, "lag_time": 10, "features_normalization": "mean_std", "input_colvars": {"start": 0, "stop": None, "stride": 1},
            "architecture": {"encoder": {"layers": [15, 15], "dropout": [0.1, 0.1]}},
            "training": {"general": {"num_tries": 2, "seed": 42, "lengths": [0.8, 0.2], "batch_size": 32, "max_epochs": 60, "shuffle": False,
                                     "random_split": False, "check_val_every_n_epoch": 10, "save_check_every_n_epoch": 10},
                         "early_stopping": {"patience": 2, "min_delta": 0.00001},
                         "optimizer": {"name": "Adam", "kwargs": {"lr": 0.01, "weight_decay": 0.0}}, "save_loss": True, "plot_loss": True}},
        "ae": {"training": {"general": {"shuffle": True, "random_split": True}, "optimizer": {"kwargs": {"lr": 0.01}}}},
        "htica": {"num_subspaces": 10, "subspaces_dimension": 5},
    }
    out = tools.train_colvars(cfg, [path], features_list=names, output_folder=str(tmp_path / "out"))
    assert sorted(out) == ["ae", "deep_tica", "pca", "tica"]
    for cv, paths in out.items():
        df = pd.read_csv(paths[0])
        assert df.shape == (164, 1) and np.all(np.isfinite(df.to_numpy())) and np.ptp(df.to_numpy()) > 0.5
        assert os.path.exists(tmp_path / "out" / cv / "model.zip")


# ----------------------------------------------------------------------------- split-K slab capacity
def test_short_slab_is_refused_not_overrun():
    """launch_gemm<kTN, EpiSlab> checks the number of split-K slabs against the capacity of the caller's buffer and
    returns DCV_ENOMEM instead of writing past it (the guard slab behind the capacity stays untouched)."""
    from deep_cartograph_amd import hip
    from deep_cartograph_amd._lib import DcvError

    A = torch.randn(4096, 96, device="cuda")
    B = torch.randn(4096, 160, device="cuda")
    slabs = hip.gemm_tn_split(A, B, k_chunk=512, slab_cap=8)
    assert torch.isnan(slabs[8]).all()                      # guard slab untouched
    ref = A.double().T @ B.double()
    np.testing.assert_allclose(slabs[:8].double().sum(0).cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-3)
    with pytest.raises(DcvError, match="slab"):
        hip.gemm_tn_split(A, B, k_chunk=512, slab_cap=7)    # 8 splits needed


# ----------------------------------------------------------------------------- tools under torch.distributed (2 ranks, 1 GPU)
def _tools_rank(rank, world, port, tmpdir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from deep_cartograph_amd import tools

    cfg = json.load(open(os.path.join(tmpdir, "cfg.json")))
    tools.train_colvars(cfg, [os.path.join(tmpdir, "a.dat"), os.path.join(tmpdir, "b.npy")], output_folder=os.path.join(tmpdir, "out2"))
    dist.destroy_process_group()


def test_train_colvars_two_ranks_match_single_process(tmp_path):
    """load_training_data shards the frames over the ranks (contiguous blocks, lag-row halo for the covariances), rank 0
    writes the CSVs and model.zip: the same files as a single process (linear CVs to 1e-4 in the '%.4f' CSV)."""
    import socket

    import torch.multiprocessing as mp

    from deep_cartograph_amd import colvars, tools

    X = ar_features(1201, 24, 41)
    names = [f"d{i}" for i in range(24)]
    colvars.write_colvars(str(tmp_path / "a.dat"), X[:500], names)
    colvars.write_binary_matrix(str(tmp_path / "b.npy"), X[500:], names)
    cfg = {"cvs": ["pca", "tica", "htica"], "common": {"dimension": 2, "lag_time": 3, "features_normalization": "mean_std",
                                                       "num_subspaces": 4, "subspaces_dimension": 3}}
    json.dump(cfg, open(tmp_path / "cfg.json", "w"))
    one = tools.train_colvars(cfg, [str(tmp_path / "a.dat"), str(tmp_path / "b.npy")], output_folder=str(tmp_path / "out1"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_tools_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for cv, paths in one.items():
        for p in paths:
            a = pd.read_csv(p).to_numpy()
            b = pd.read_csv(p.replace("out1", "out2")).to_numpy()
            assert a.shape == b.shape
            np.testing.assert_allclose(a, b, atol=2e-4)
        assert os.path.exists(tmp_path / "out2" / cv / "model.zip")
