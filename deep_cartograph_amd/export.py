"""TorchScript export / import of the neural CVs with the module tree the reference writes to
``model/cv_weights.pt``: ``DeepTICA`` = ``norm_in -> nn -> tica ->
postprocessing``, ``AutoEncoderCV`` = ``norm_in -> encoder -> postprocessing`` (decoder
parameters present but unused in ``forward``), ``FeedForward.nn`` a ``Sequential`` of
Linear / activation / Dropout, parameter names ``nn.nn.{i}.weight`` / ``encoder.nn.{i}.weight``.
PLUMED's ``PYTORCH_MODEL`` consumes the file directly.  torch is export glue here: the numbers
inside come from the HIP engine."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

class Shifted_Softplus(torch.nn.Softplus):
    """mlcolvar.core.nn.utils.Shifted_Softplus: softplus shifted to pass through the origin."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.softplus(input, self.beta, self.threshold) - 0.6931471824645996


class Custom_Sigmoid(torch.nn.Module):
    """mlcolvar.core.nn.utils.Custom_Sigmoid: 1 / (1 + exp(-p x)), p = 3."""

    def __init__(self, p: float = 3.0):
        super().__init__()
        self.p = p

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return 1 / (1 + torch.exp(-self.p * input))


_ACT_MODULES = {
    "shifted_softplus": lambda: Shifted_Softplus(),
    "custom_sigmoid": lambda: Custom_Sigmoid(),
    "relu": lambda: torch.nn.ReLU(True),
    "elu": lambda: torch.nn.ELU(True),
    "tanh": lambda: torch.nn.Tanh(),
    "softplus": lambda: torch.nn.Softplus(),
    "leaky_relu": lambda: torch.nn.LeakyReLU(),
}
_ACT_FROM_NAME = {"Shifted_Softplus": "shifted_softplus", "Custom_Sigmoid": "custom_sigmoid", "ReLU": "relu", "ELU": "elu", "Tanh": "tanh", "Softplus": "softplus", "LeakyReLU": "leaky_relu"}


class Normalization(torch.nn.Module):
    def __init__(self, mean, rng):
        super().__init__()
        self.register_buffer("mean", torch.as_tensor(np.asarray(mean), dtype=torch.float32).clone())
        self.register_buffer("range", torch.as_tensor(np.asarray(rng), dtype=torch.float32).clone())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.sub(self.mean.expand_as(x)).div(self.range.expand_as(x))


class FeedForward(torch.nn.Module):
    """Linear [, activation][, Dropout][, BatchNorm1d] per layer, in the reference's module order (mlcolvar FeedForward).
    ``batchnorm``: per Linear None or the dict of its BatchNorm1d (weight, bias, running_mean, running_var,
    num_batches_tracked)."""

    def __init__(self, linears: Sequence[Tuple[np.ndarray, np.ndarray]], activation: Sequence[Optional[str]],
                 dropout: Optional[Sequence[Optional[float]]] = None, batchnorm: Optional[Sequence[Optional[dict]]] = None):
        super().__init__()
        mods: List[torch.nn.Module] = []
        for i, (w, b) in enumerate(linears):
            lin = torch.nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(torch.as_tensor(w, dtype=torch.float32))
                lin.bias.copy_(torch.as_tensor(b, dtype=torch.float32))
            mods.append(lin)
            act = activation[i]
            if act not in (None, "linear"):
                mods.append(_ACT_MODULES[act]())
            if dropout is not None and dropout[i] is not None:
                mods.append(torch.nn.Dropout(p=float(dropout[i])))
            if batchnorm is not None and batchnorm[i] is not None:
                st = batchnorm[i]
                bn = torch.nn.BatchNorm1d(w.shape[0])
                with torch.no_grad():
                    bn.weight.copy_(torch.as_tensor(st["weight"], dtype=torch.float32))
                    bn.bias.copy_(torch.as_tensor(st["bias"], dtype=torch.float32))
                    bn.running_mean.copy_(torch.as_tensor(st["running_mean"], dtype=torch.float32))
                    bn.running_var.copy_(torch.as_tensor(st["running_var"], dtype=torch.float32))
                    bn.num_batches_tracked.fill_(int(st.get("num_batches_tracked", 0)))
                mods.append(bn)
        self.nn = torch.nn.Sequential(*mods)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.nn(x)


class TICA(torch.nn.Module):
    def __init__(self, evecs, mean):
        super().__init__()
        self.register_buffer("evecs", torch.as_tensor(np.asarray(evecs), dtype=torch.float32).clone())
        self.register_buffer("mean", torch.as_tensor(np.asarray(mean), dtype=torch.float32).clone())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.matmul(x.sub(self.mean), self.evecs)


class ReduceEigenvaluesLoss(torch.nn.Module):
    def forward(self, evals: torch.Tensor) -> torch.Tensor:
        return -torch.sum(torch.pow(evals, 2))


class MSELoss(torch.nn.Module):
    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return (a - b).square().mean()


class DeepTICA(torch.nn.Module):
    def __init__(self, norm_in, nn, tica, postprocessing):
        super().__init__()
        self.loss_fn = ReduceEigenvaluesLoss()
        self.norm_in = norm_in
        self.nn = nn
        self.tica = tica
        self.postprocessing = postprocessing

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.norm_in is not None:
            x = self.norm_in(x)
        x = self.tica(self.nn(x))
        if self.postprocessing is not None:
            x = self.postprocessing(x)
        return x


class AutoEncoderCV(torch.nn.Module):
    def __init__(self, norm_in, encoder, decoder, postprocessing):
        super().__init__()
        self.loss_fn = MSELoss()
        self.norm_in = norm_in
        self.encoder = encoder
        self.decoder = decoder
        self.postprocessing = postprocessing

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.norm_in is not None:
            x = self.norm_in(x)
        x = self.encoder(x)
        if self.postprocessing is not None:
            x = self.postprocessing(x)
        return x


def save_torchscript(model: torch.nn.Module, n_features: int, path: str) -> None:
    """to_torchscript(method='trace') in eval mode, as NonLinear.save_weights does
    (cv_calculator.py:1773-1795)."""
    model.eval()
    example = torch.zeros(2, n_features, dtype=torch.float32)
    with torch.no_grad():
        traced = torch.jit.trace(model, example)
    traced.save(path)


def _sequential_layers(seq):
    """(linears, activation per linear, batch normalisation per linear) from a scripted Sequential of
    Linear / act / Dropout / BatchNorm1d."""
    linears, acts, bns = [], [], []
    for child in seq.children():
        name = getattr(child, "original_name", type(child).__name__)
        if name == "Linear":
            linears.append((child.weight.detach().cpu().numpy().copy(), child.bias.detach().cpu().numpy().copy()))
            acts.append(None)
            bns.append(None)
        elif name in _ACT_FROM_NAME:
            acts[-1] = _ACT_FROM_NAME[name]
        elif name in ("Dropout", "Identity"):
            continue
        elif name == "BatchNorm1d":
            st = {k: v.detach().cpu().numpy().copy() for k, v in list(child.named_parameters()) + list(child.named_buffers())}
            bns[-1] = {"weight": st["weight"], "bias": st["bias"], "running_mean": st["running_mean"], "running_var": st["running_var"],
                       "num_batches_tracked": int(st.get("num_batches_tracked", 0))}
        else:
            raise ValueError(f"TorchScript layer {name} is not supported by the HIP engine")
    return linears, acts, bns


def read_torchscript(path: str) -> dict:
    """Decompose a reference-format cv_weights.pt into arrays the HIP engine can run:
    kind ('deep_tica' | 'ae'), linears, acts, norm_in (mean, range) or None, tica (mean, evecs)
    or None, postprocessing (mean, range) or None."""
    m = torch.jit.load(path, map_location="cpu")
    m.eval()
    kids = dict(m.named_children())
    buf = {k: v.detach().cpu().numpy().copy() for k, v in m.named_buffers()}

    def pair(prefix):
        if f"{prefix}.mean" in buf and f"{prefix}.range" in buf:
            return buf[f"{prefix}.mean"], buf[f"{prefix}.range"]
        return None

    out = {"norm_in": pair("norm_in"), "postprocessing": pair("postprocessing"), "tica": None, "module": m}
    if "nn" in kids and "tica" in kids:
        out["kind"] = "deep_tica"
        out["linears"], out["acts"], out["bn"] = _sequential_layers(kids["nn"].nn)
        out["tica"] = (buf["tica.mean"], buf["tica.evecs"])
    elif "encoder" in kids:
        out["kind"] = "ae"
        out["linears"], out["acts"], out["bn"] = _sequential_layers(kids["encoder"].nn)
    else:
        raise ValueError("unrecognised TorchScript CV model (expected a DeepTICA or AutoEncoderCV tree)")
    return out
