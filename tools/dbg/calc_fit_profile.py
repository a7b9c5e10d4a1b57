"""Developer probe: cProfile of a calculator fit (where the host time of an epoch goes)."""
import cProfile, json, pstats, sys, tempfile
import torch
sys.path.insert(0, ".")
from tests.test_mlp_gpu import ar_features
from tests.test_calculators_gpu import TEST_COMMON
from deep_cartograph_amd.cv_calculator import cv_calculators_map

kind = sys.argv[1]
n, epochs, bs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X = ar_features(n, 54, 3)
cfg = json.loads(json.dumps(TEST_COMMON))
cfg["training"]["general"].update({"batch_size": bs, "max_epochs": epochs, "shuffle": True, "random_split": True})
cfg["training"]["early_stopping"]["patience"] = 10000
cfg["lag_time"] = 5
with tempfile.TemporaryDirectory() as out:
    calc = cv_calculators_map[kind](cfg, out)
    calc.set_training_matrix(X.copy(), [f"f{i}" for i in range(54)])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    calc.train()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
