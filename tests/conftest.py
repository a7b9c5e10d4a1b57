import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def features():
    g = load_golden("features_164x54.npz")
    return np.ascontiguousarray(g["X"]), [str(s) for s in g["names"]]


@pytest.fixture(scope="session")
def golden_proj():
    return load_golden("train_colvars_golden.npz")


@pytest.fixture(scope="session")
def golden_linear():
    return load_golden("linear_models.npz")


@pytest.fixture(scope="session")
def golden_nn():
    return load_golden("nn_models.npz")


@pytest.fixture(scope="session")
def golden_cluster():
    return load_golden("cluster_golden.npz")
