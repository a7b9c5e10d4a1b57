"""Developer probe: a validation pass batch by batch (dcv_mlp_eval_step) against the one-call form (dcv_mlp_eval_steps)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip

def run(model, dims, bs, nb, reps=20):
    n = bs * nb + 64
    X = torch.randn(n, dims[0], device="cuda")
    if model == "ae":
        full = dims + dims[-2::-1]
        acts = (["tanh"] * (len(dims) - 2) + [None]) * 2
        eng = hip.Mlp("ae", full, acts, max_batch=bs, latent_layer=len(dims) - 1)
        eng.set_feature_range(np.ones(dims[0], dtype=np.float32))
    else:
        full = dims
        acts = ["tanh"] * (len(dims) - 2) + [None]
        eng = hip.Mlp("deep_tica", dims, acts, max_batch=bs, lag=5, tica_reg=1e-6)
    torch.manual_seed(0)
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in [torch.nn.Linear(full[i], full[i + 1]) for i in range(len(full) - 1)]])
    eng.reset_log(2 * nb * (reps + 2))
    def loop():
        for j in range(nb):
            eng.eval_step(X, row0=j * bs, batch=bs)
    def one():
        eng.eval_steps(X, bs, nb, row0=0)
    for f, name in ((loop, "step by step"), (one, "one call")):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        print(f"{model} {'-'.join(map(str, dims))} batch {bs} x {nb}: {name}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per pass (path {eng.last_path()})")
        eng.reset_log(2 * nb * (reps + 2))

run("ae", [128, 64, 32, 2], 4096, 48)
run("ae", [54, 16, 8, 2], 128, 40)
run("deep_tica", [54, 16, 8, 2], 128, 40)
run("deep_tica", [54, 15, 15, 2], 4096, 20)
