"""CPU-side checks: the C-ABI library builds, loads and exports every symbol of include/dcv.h;
the product path refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "dcv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    import __graft_entry__ as ge
    from deep_cartograph_amd import _lib

    ge.build()
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dcv.h but not exported by libdcv.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert lib.dcv_abi_version() == 2


def test_no_cpu_fallback():
    from deep_cartograph_amd import hip
    from deep_cartograph_amd._lib import DcvError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    X = torch.zeros(8, 4)
    with pytest.raises(DcvError):
        hip.col_stats_raw(X)
    with pytest.raises(DcvError):
        hip.Mlp("deep_tica", [4, 2], [None], max_batch=8)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "deep_cartograph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
