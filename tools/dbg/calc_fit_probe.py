"""Developer probe: wall time of a user-scale calculator fit (the whole host path around the kernels): n frames x 54 features,
the reference's default architecture, batch 256, shuffled loader, E epochs -- time per epoch and per training step."""
import json, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_mlp_gpu import ar_features
from tests.test_calculators_gpu import TEST_COMMON
from deep_cartograph_amd.cv_calculator import cv_calculators_map

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 256
X = ar_features(n, 54, 3)
names = [f"f{i}" for i in range(54)]
for kind in ("deep_tica", "ae"):
    cfg = json.loads(json.dumps(TEST_COMMON))
    cfg["training"]["general"].update({"batch_size": bs, "max_epochs": epochs, "shuffle": True, "random_split": True})
    cfg["training"]["early_stopping"]["patience"] = 10000
    cfg["lag_time"] = 5
    with tempfile.TemporaryDirectory() as out:
        calc = cv_calculators_map[kind](cfg, out)
        calc.set_training_matrix(X.copy(), names)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = calc.train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = epochs * (int(n * 0.8) // bs)
        print(f"{kind}: n={n} batch={bs} epochs={epochs}: train() {dt:.3f} s = {dt / epochs * 1e3:.2f} ms / epoch, {dt / steps * 1e6:.1f} us per training step "
              f"(incl. validation, logging, snapshots); {steps * bs / dt / 1e6:.2f} M frames/s; ok={ok}")
