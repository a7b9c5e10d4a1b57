"""Developer probe: fused Deep-TICA training step time (54-16-8-2) over batch sizes, for the tile rule in the environment."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip
dims = [54, 16, 8, 2]
out = []
for bs in (64, 128, 256, 512, 1024, 2048):
    n = bs * 40 + 64
    X = torch.randn(n, 54, device="cuda")
    eng = hip.Mlp("deep_tica", dims, ["leaky_relu", "leaky_relu", None], max_batch=bs, lag=5, tica_reg=1e-6, lr=1e-3)
    torch.manual_seed(0)
    eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]])
    eng.reset_log(40 * 60)
    for _ in range(5): eng.train_steps(X, bs, 40, row0=0)
    eng.reset_log(40 * 60)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(25): eng.train_steps(X, bs, 40, row0=0)
    torch.cuda.synchronize()
    out.append("%d:%.1f" % (bs, (time.perf_counter() - t0) / 1000 * 1e6))
print("DT16=%s  us/step  %s" % (os.environ.get("DCV_SNET_DT16", "-"), "  ".join(out)))
