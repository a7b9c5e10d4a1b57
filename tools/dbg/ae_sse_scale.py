"""How the autoencoder loss pass scales with batch and width (run under rocprofv3 --kernel-trace --stats)."""
import sys
import torch
sys.path.insert(0, ".")
from deep_cartograph_amd import hip

for F in (64, 128, 512):
    X = torch.randn(40000, F, device="cuda")
    for batch in (1024, 4096, 16384):
        dims = [F, 32, 2, 32, F]
        eng = hip.Mlp("ae", dims, ["tanh", None, "tanh", None], max_batch=batch, latent_layer=2)
        eng.reset_log(64)
        for _ in range(20):
            eng.eval_step(X, row0=0, batch=batch)
        torch.cuda.synchronize()
        eng.close()
