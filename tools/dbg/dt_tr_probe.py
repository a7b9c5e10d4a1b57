"""Developer probe: fused Deep-TICA training step time (54-16-8-2) over batch sizes, for the tile rule in the environment."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip
for dims in ([54, 16, 8, 2], [128, 64, 32, 4]):
  out = []
  for bs in (128, 512, 1024, 2048, 4096):
      n = bs * 40 + 64
      X = torch.randn(n, dims[0], device="cuda")
      eng = hip.Mlp("deep_tica", dims, ["leaky_relu"] * (len(dims) - 2) + [None], max_batch=bs, lag=5, tica_reg=1e-6, lr=1e-3)
      torch.manual_seed(0)
      eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)]])
      eng.reset_log(40 * 60)
      for _ in range(5): eng.train_steps(X, bs, 40, row0=0)
      eng.reset_log(40 * 60)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(25): eng.train_steps(X, bs, 40, row0=0)
      torch.cuda.synchronize()
      out.append("%d:%.1f" % (bs, (time.perf_counter() - t0) / 1000 * 1e6))
  print("TR=%s  us/step  %s" % (os.environ.get("DCV_SNET_TR", "rule") + " " + "-".join(map(str, dims)), "  ".join(out)))
