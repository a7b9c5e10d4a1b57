"""Oracle: AE / Deep-TICA models, training loop, post-normalisation and forward pass.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain torch-CPU autograd restatement of what
mlcolvar 1.2.2 + lightning 2.5.1 execute for cv_calculator.py:1456-1553 (NonLinear.train),
:2471-2492 (AutoEncoderCV), :2569-2590 (DeepTICA) and :1735-1754 (normalize_cv).
The third-party behaviour restated here is listed in SURVEY.md Appendix A.5/A.6/A.9.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from .linear import cholesky_eigh, correlation_matrix

class ShiftedSoftplus(torch.nn.Softplus):
    """mlcolvar Shifted_Softplus [from knowledge of mlcolvar 1.2.2, core/nn/utils.py]: softplus(x) - softplus(0)."""

    def forward(self, x):
        sp0 = torch.nn.functional.softplus(torch.zeros(1), self.beta, self.threshold).item()
        return torch.nn.functional.softplus(x, self.beta, self.threshold) - sp0


class CustomSigmoid(torch.nn.Module):
    """mlcolvar Custom_Sigmoid [from knowledge]: 1 / (1 + exp(-p x)), p = 3."""

    def __init__(self, p=3):
        super().__init__()
        self.p = p

    def forward(self, x):
        return 1 / (1 + torch.exp(-self.p * x))


class MaskedDropout(torch.nn.Module):
    """torch.nn.Dropout whose keep / (1 - p) multipliers can be injected (tests hand it the masks the HIP engine
    drew, since the reference's CPU generator stream cannot be reproduced on the device): `queue` holds one
    multiplier tensor per forward call in training mode; empty queue -> ordinary torch dropout."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)
        self.queue = []

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.queue:
            return x * self.queue.pop(0)
        return torch.nn.functional.dropout(x, self.p, True)


ACTIVATIONS = {
    "shifted_softplus": lambda: ShiftedSoftplus(),
    "custom_sigmoid": lambda: CustomSigmoid(),
    "relu": lambda: torch.nn.ReLU(True),
    "elu": lambda: torch.nn.ELU(True),
    "tanh": lambda: torch.nn.Tanh(),
    "softplus": lambda: torch.nn.Softplus(),
    "leaky_relu": lambda: torch.nn.LeakyReLU(),  # slope 0.01 (Appendix A.5)
    "linear": None,
    None: None,
}


def feed_forward(layers: Sequence[int], activation: Sequence, dropout: Optional[Sequence] = None,
                 batchnorm: Optional[Sequence] = None) -> torch.nn.Sequential:
    """mlcolvar.core.nn.FeedForward body: Linear [, act][, Dropout][, BatchNorm1d] per layer (Appendix A.5; the order
    activation -> dropout -> batchnorm is restated from knowledge of mlcolvar 1.2.2's feedforward.py -- parity unpinned:
    no fixture of the reference carries a batch normalisation).
    ``activation`` / ``dropout`` have one entry per Linear (the reference appends the
    last-layer entry itself, cv_calculator.py:1155-1219).  Consumes the global torch RNG in
    construction order exactly as nn.Linear does."""
    n = len(layers) - 1
    assert len(activation) == n
    if dropout is None:
        dropout = [None] * n
    mods: List[torch.nn.Module] = []
    for i in range(n):
        mods.append(torch.nn.Linear(layers[i], layers[i + 1]))
        act = activation[i]
        if ACTIVATIONS[act] is not None:
            mods.append(ACTIVATIONS[act]())
        if dropout[i] is not None:
            mods.append(MaskedDropout(p=dropout[i]))
        if batchnorm is not None and batchnorm[i]:
            mods.append(torch.nn.BatchNorm1d(layers[i + 1]))
    return torch.nn.Sequential(*mods)


class Normalization(torch.nn.Module):
    """mlcolvar.core.transform.Normalization: (x - mean) / range; inverse x * range + mean."""

    def __init__(self, mean, rng):
        super().__init__()
        self.register_buffer("mean", torch.as_tensor(mean, dtype=torch.float32).clone())
        self.register_buffer("range", torch.as_tensor(rng, dtype=torch.float32).clone())

    def forward(self, x):
        return x.sub(self.mean).div(self.range)

    def inverse(self, x):
        return x.mul(self.range).add(self.mean)


def minmax_normalization(Y: torch.Tensor) -> Normalization:
    """Normalization(mode='min_max', stats=Statistics(Y)): mean=(max+min)/2, range=(max-min)/2
    (cv_calculator.py:1750-1754; Appendix A.1/a12)."""
    mn = Y.min(dim=0).values
    mx = Y.max(dim=0).values
    return Normalization((mx + mn) / 2.0, (mx - mn) / 2.0)


# ----------------------------------------------------------------------------- Deep-TICA
def batch_tica(f_t: torch.Tensor, f_lag: torch.Tensor, reg: float):
    """TICA.compute on one batch, unit weights, remove_average=True (Appendix A.2).
    Returns (evals, evecs, mean); differentiable."""
    mu = f_t.mean(dim=0)
    xc = f_t - mu
    yc = f_lag - mu
    C0 = correlation_matrix(xc, xc)
    Ct = correlation_matrix(xc, yc)
    evals, evecs = cholesky_eigh(Ct, C0, reg, n_eig=0)
    return evals, evecs, mu


def deeptica_loss(evals: torch.Tensor) -> torch.Tensor:
    """ReduceEigenvaluesLoss(mode='sum2'): -sum(lambda_i^2) (verified)."""
    return -torch.sum(evals.pow(2))


class DeepTICAModel(torch.nn.Module):
    """norm_in -> nn -> tica (-> postprocessing); module tree of Appendix A.5."""

    def __init__(self, layers, activation, dropout, norm_mean, norm_range, reg, batchnorm=None):
        super().__init__()
        self.norm_in = Normalization(norm_mean, norm_range) if norm_mean is not None else None
        self.nn = feed_forward(layers, activation, dropout, batchnorm)
        d = layers[-1]
        self.reg = reg
        self.register_buffer("tica_evecs", torch.eye(d))
        self.register_buffer("tica_mean", torch.zeros(d))
        self.register_buffer("tica_evals", torch.zeros(d))
        self.postprocessing: Optional[Normalization] = None

    def forward_nn(self, x):
        if self.norm_in is not None:
            x = self.norm_in(x)
        return self.nn(x)

    def step(self, x_t, x_lag):
        f_t = self.forward_nn(x_t)
        f_lag = self.forward_nn(x_lag)
        evals, evecs, mu = batch_tica(f_t, f_lag, self.reg)
        # save_params=True: every train AND validation step overwrites the buffers (A.6 ii)
        self.tica_evals = evals.detach().clone()
        self.tica_evecs = evecs.detach().clone()
        self.tica_mean = mu.detach().clone()
        return deeptica_loss(evals), evals.detach()

    def forward(self, x):
        y = (self.forward_nn(x) - self.tica_mean) @ self.tica_evecs
        if self.postprocessing is not None:
            y = self.postprocessing(y)
        return y


# ----------------------------------------------------------------------------- AE
class AEModel(torch.nn.Module):
    """AutoEncoderCV: loss = mean((norm_in.inverse(decoder(encoder(norm_in(x)))) - x)^2)
    over batch x features (Appendix A.9)."""

    def __init__(self, enc_layers, enc_act, enc_drop, dec_layers, dec_act, dec_drop, norm_mean, norm_range, enc_bn=None, dec_bn=None):
        super().__init__()
        self.norm_in = Normalization(norm_mean, norm_range) if norm_mean is not None else None
        self.encoder = feed_forward(enc_layers, enc_act, enc_drop, enc_bn)
        self.decoder = feed_forward(dec_layers, dec_act, dec_drop, dec_bn)
        self.postprocessing: Optional[Normalization] = None

    def forward_cv(self, x):
        if self.norm_in is not None:
            x = self.norm_in(x)
        return self.encoder(x)

    def step(self, x):
        x_hat = self.decoder(self.forward_cv(x))
        if self.norm_in is not None:
            x_hat = self.norm_in.inverse(x_hat)
        diff = x_hat - x
        return (diff * diff).mean(), None

    def forward(self, x):
        y = self.forward_cv(x)
        if self.postprocessing is not None:
            y = self.postprocessing(y)
        return y


# ----------------------------------------------------------------------------- data plan
def split_indices(n: int, lengths: Sequence[float], random_split: bool, generator=None):
    """DictModule split.  random_split=True -> torch.utils.data.random_split with fractional
    lengths (floor + round-robin remainder, one randperm from ``generator``); False ->
    sequential blocks (Appendix A.6 / A.9)."""
    sizes = [int(math.floor(n * f)) for f in lengths]
    rem = n - sum(sizes)
    for i in range(rem):
        sizes[i % len(sizes)] += 1
    if random_split:
        perm = torch.randperm(n, generator=generator)
    else:
        perm = torch.arange(n)
    out, off = [], 0
    for s in sizes:
        out.append(perm[off: off + s].clone())
        off += s
    return out


def batches(idx: torch.Tensor, batch_size: int, shuffle: bool):
    """DictLoader: consecutive slices of ``batch_size`` (last partial batch kept); a fresh
    randperm per epoch when shuffling; batch_size 0 -> one batch (Appendix A.9)."""
    n = len(idx)
    if shuffle:
        idx = idx[torch.randperm(n)]
    if batch_size <= 0 or batch_size >= n:
        return [idx]
    return [idx[i: i + batch_size] for i in range(0, n, batch_size)]


def closest_power_of_two(n: int) -> int:
    """modules/common/common.py:645-666 -- 2**floor(log2 n), halved when n is itself a power of two
    (strictly below n); n = 1 (reference: 0.5) is clamped to 1."""
    if n <= 1:
        return 1
    p = 2 ** math.floor(math.log2(n))
    return p // 2 if p == n else p


def clamp_batch_size(batch_size: int, n_total: int, train_frac: float) -> int:
    """check_num_samples + check_batch_size (cv_calculator.py:1278-1309): the estimate
    int(n_total * frac) is used only for the clamp."""
    n_train = int(n_total * train_frac)
    if batch_size >= n_train:
        return closest_power_of_two(n_train)
    return batch_size


# ----------------------------------------------------------------------------- trainer
def train(model, data: dict, *, seed_try: int, lengths=(0.8, 0.2), batch_size=32, shuffle=False,
          random_split=True, max_epochs=100, check_val_every_n_epoch=1, save_check_every_n_epoch=1,
          patience=20, min_delta=1e-5, optimizer="Adam", opt_kwargs=None, model_to_save="best",
          build_model=None, val_data: Optional[dict] = None, scheduler: Optional[dict] = None):
    """One training try of NonLinear.train (cv_calculator.py:1478-1539) without lightning.

    ``data``: {'data': X} for AE or {'data': x_t, 'data_lag': x_lag} for Deep-TICA (CPU f32).
    ``val_data``: the same for a separately supplied validation set -- then the whole of ``data`` is
    trained on and nothing is split (cv_calculator.py:1485-1492).
    ``scheduler``: {'name', 'kwargs', 'config': {'interval', 'frequency', 'monitor'}} after
    adjust_lr_scheduler (:1228-1273); stepped as lightning does (interval 'step': after every optimiser
    step; 'epoch': at the end of every epoch, ReduceLROnPlateau with the latest valid_loss).
    RNG order (Appendix A.6): manual_seed(seed_try) -> model construction (``build_model()``) -> randperm
    for the split; with a scheduler and no separate validation set the split comes first (:1503 vs :1507).
    Pass ``model=None, build_model=fn`` to have construction happen at the right point of the RNG stream.
    Returns dict(model, metrics, score, split)."""
    import copy

    opt_kwargs = dict(opt_kwargs or {"lr": 1e-3})
    gen = torch.manual_seed(seed_try)
    n = data["data"].shape[0]
    split = None
    if scheduler is not None and val_data is None:
        split = split_indices(n, lengths, random_split, gen)
    if model is None:
        model = build_model()
    if val_data is not None:
        train_idx, val_idx = torch.arange(n), torch.arange(val_data["data"].shape[0])
    else:
        train_idx, val_idx = split if split is not None else split_indices(n, lengths, random_split, gen)
        val_data = data
    opt = getattr(torch.optim, optimizer)(model.parameters(), **opt_kwargs)
    is_tica = "data_lag" in data
    sched, s_interval, s_freq, s_count = None, "epoch", 1, 0
    if scheduler is not None:
        sched = getattr(torch.optim.lr_scheduler, scheduler["name"])(opt, **scheduler.get("kwargs", {}))
        s_interval = scheduler.get("config", {}).get("interval", "epoch")
        s_freq = max(1, int(scheduler.get("config", {}).get("frequency", 1)))
    plateau = isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau)

    def sched_step(metric=None):
        nonlocal s_count
        s_count += 1
        if s_count % s_freq:
            return
        if plateau:
            if metric is None:
                raise ValueError("ReduceLROnPlateau conditioned on a metric which is not available yet")
            sched.step(metric)
        else:
            sched.step()

    def run(src, idx):
        if is_tica:
            return model.step(src["data"][idx], src["data_lag"][idx])
        return model.step(src["data"][idx])

    metrics = {"train_loss": [], "valid_loss": [], "epoch": []}
    best_score, best_state, wait = float("inf"), None, 0
    es_best = float("inf")
    last_state, last_score = None, None
    last_valid = None
    for epoch in range(max_epochs):
        model.train()
        tot, cnt = 0.0, 0
        for b in batches(train_idx, batch_size, shuffle):
            opt.zero_grad()
            loss, _ = run(data, b)
            loss.backward()
            opt.step()
            if sched is not None and s_interval == "step":
                sched_step()
            tot += float(loss.detach()) * len(b)
            cnt += len(b)
        train_loss = tot / cnt
        stop = False
        if (epoch + 1) % check_val_every_n_epoch == 0:
            model.eval()
            vt, vc, eig = 0.0, 0, None
            with torch.no_grad():
                for b in batches(val_idx, batch_size, shuffle):
                    loss, ev = run(val_data, b)
                    vt += float(loss) * len(b)
                    vc += len(b)
                    if ev is not None:
                        eig = (ev * len(b)) if eig is None else eig + ev * len(b)
            valid_loss = vt / vc
            last_valid = valid_loss
            metrics["train_loss"].append(train_loss)
            metrics["valid_loss"].append(valid_loss)
            metrics["epoch"].append(epoch)
            if eig is not None:
                for i, v in enumerate((eig / vc).tolist()):
                    metrics.setdefault(f"valid_eigval_{i + 1}", []).append(v)
            # ModelCheckpoint (save_top_k=1, save_last=True, every_n_epochs)
            if (epoch + 1) % save_check_every_n_epoch == 0:
                last_state, last_score = copy.deepcopy(model.state_dict()), valid_loss
                if valid_loss < best_score:
                    best_score, best_state = valid_loss, copy.deepcopy(model.state_dict())
            # EarlyStopping(monitor=valid_loss, mode=min)
            if valid_loss < es_best - min_delta:
                es_best, wait = valid_loss, 0
            else:
                wait += 1
                stop = wait >= patience
        if sched is not None and s_interval == "epoch":
            sched_step(last_valid)
        if stop:
            break
    if model_to_save == "best" and best_state is not None:
        model.load_state_dict(best_state)
        score = best_score
    else:
        if last_state is not None:
            model.load_state_dict(last_state)
        # _finalize_training (cv_calculator.py:1566): last_score = metrics['valid_loss'][-1], the FINAL validation loss
        score = metrics["valid_loss"][-1] if metrics["valid_loss"] else last_score
    model.eval()
    return {"model": model, "metrics": metrics, "score": score, "split": (train_idx, val_idx)}


def finalize_postprocessing(model, X_rows: torch.Tensor):
    """NonLinear.normalize_cv (cv_calculator.py:1735-1754): forward the training rows with
    postprocessing=None, min/max per CV, attach Normalization(mode='min_max')."""
    model.postprocessing = None
    with torch.no_grad():
        Y = model(X_rows)
    model.postprocessing = minmax_normalization(Y)
    return model


def sensitivity_mean_abs(model, X_rows: torch.Tensor, std=None):
    """mlcolvar.explain.sensitivity_analysis(model, dataset, metric="mean_abs_val") as called by
    NonLinear.sensitivity_analysis (cv_calculator.py:1903), restated from the published algorithm of
    mlcolvar 1.2.2 (the package is not under /root/reference and no reference fixture holds its output:
    PARITY UNPINNED for this function): gradient of the summed outputs with respect to the raw inputs
    (grad_outputs = ones), multiplied by the per-feature standard deviation of dataset['data'] (torch.std,
    unbiased), mean absolute value over the samples, normalised to sum to one.  Returns the scores in
    feature order (the reference then sorts them ascending for the CSV)."""
    X = X_rows.clone().requires_grad_(True)
    out = model(X)
    grad = torch.autograd.grad(out, X, grad_outputs=torch.ones_like(out))[0].detach().double().numpy()
    if std is None:
        std = torch.std(X_rows.double(), dim=0).numpy()
    score = np.abs(grad * np.asarray(std, dtype=np.float64)).mean(axis=0)
    return score / score.sum()
