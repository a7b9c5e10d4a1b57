"""Developer probe: fused autoencoder training step time over batch sizes for two networks, tile rows forced or by the rule."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip
for enc in ([54, 16, 8, 2], [128, 64, 32, 2]):
    full = enc + enc[-2::-1]
    acts = (["leaky_relu"] * (len(enc) - 2) + [None]) * 2
    out = []
    for bs in (128, 256, 512, 1024, 2048, 4096):
        n = bs * 40 + 64
        X = torch.randn(n, enc[0], device="cuda")
        eng = hip.Mlp("ae", full, acts, max_batch=bs, latent_layer=len(enc) - 1, lr=1e-3)
        eng.set_feature_range(np.ones(enc[0], dtype=np.float32))
        torch.manual_seed(0)
        eng.set_linears([(l.weight.detach().numpy(), l.bias.detach().numpy()) for l in [torch.nn.Linear(full[i], full[i + 1]) for i in range(len(full) - 1)]])
        eng.reset_log(40 * 60)
        for _ in range(5): eng.train_steps(X, bs, 40, row0=0)
        eng.reset_log(40 * 60)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(25): eng.train_steps(X, bs, 40, row0=0)
        torch.cuda.synchronize()
        out.append("%d:%.1f" % (bs, (time.perf_counter() - t0) / 1000 * 1e6))
    print("TR=%s %s  us/step  %s" % (os.environ.get("DCV_SNET_TR", "rule"), "-".join(map(str, enc)), "  ".join(out)))
