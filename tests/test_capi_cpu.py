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
    assert lib.dcv_abi_version() == 5


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


# ------------------------------------------------------------------ in-launch hand-offs: the emitted instruction order
_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _device_disassembly(tmp_path):
    """Disassembly of every gfx950 code object inside the built libdcv.so: {kernel name: [instruction lines]}."""
    import shutil
    import subprocess

    import __graft_entry__ as ge
    from deep_cartograph_amd import _lib

    ge.build()
    so = os.path.join(str(tmp_path), "libdcv.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([_OBJDUMP, "--offloading", so], check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    kernels = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        text = subprocess.run([_OBJDUMP, "-d", os.path.join(str(tmp_path), f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]{16} <([^>]+)>:", line)
            if m:
                name = m.group(1)
                kernels[name] = []
            elif name is not None and line.startswith("\t"):
                kernels[name].append(line.split("//")[0].strip())
    return kernels


@pytest.mark.skipif(not os.path.exists(_OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_handoff_isa_order(tmp_path):
    """Every ticketed hand-off between workgroups (csrc/handoff.h: the contraction-split tail tile of the row-tiled
    products, the Deep-TICA batch statistics + loss head, the autoencoder SSE) must be emitted in the order
    MI355X_MICROARCH.md validates: write-through (sc1) payload stores -> s_waitcnt vmcnt(0) in the storing wave ->
    s_barrier -> the ticket (a returning global_atomic_add) -> in the last arriver buffer_inv sc1 + s_waitcnt vmcnt(0)
    -> s_barrier -> the loads.  Checked on the code that ships: the gfx950 code objects inside libdcv.so.
    (Round 2's form -- a workgroup-scope release -- compiled to store; s_barrier; atomic with no wait in between.)"""
    kernels = _device_disassembly(tmp_path)
    assert len(kernels) > 200
    is_store = lambda s: s.startswith(("global_store", "buffer_store", "flat_store", "scratch_store"))
    ticketed, elections = {}, {}
    for name, ins in kernels.items():
        for i, s in enumerate(ins):
            if not (re.match(r"global_atomic_add\s", s) and " sc0" in s):   # a returning add = a ticket
                continue
            if not any(x.startswith("global_load") and " sc1" in x for x in ins[i + 1:]):
                # An ELECTION, not a hand-off: the batched validation pass of the fused small-network kernels ends every batch
                # on a second ticket that only picks the workgroup that moves the log counter by the number of batches.  Nothing
                # is handed over inside the launch behind it (the records are read by later launches / the host), so there is
                # no payload to publish and nothing to acquire; what it needs is that each taker has READ the counter first:
                # an s_waitcnt vmcnt(0) between the last load ahead of it and the add.
                assert re.search(r"snet_ae_kernel|snet_dt_fwd_kernel", name), f"{name}: a ticket with no sc1 load behind it"
                j = i - 1
                while j >= 0 and not ins[j].startswith(("global_load", "flat_load", "buffer_load")):
                    j -= 1
                assert any(x.startswith("s_waitcnt") and "vmcnt(0)" in x for x in ins[j:i]), f"{name}: the election ticket does not wait for the counter read"
                elections[name] = elections.get(name, 0) + 1
                continue
            # ---- producer side: ... stores ; s_waitcnt vmcnt(0) ; s_barrier ; (no store) ; ticket -- and the payload
            # (the partials handed over in-launch) written with sc1 stores somewhere ahead of that wait.  Plain stores
            # ahead of the wait are outputs for LATER launches (gradient partials of the fused small-network step).
            j = i - 1
            while j >= 0 and not ins[j].startswith("s_barrier"):
                assert not is_store(ins[j]), f"{name}: a store sits between the barrier and the ticket: {ins[j]}"
                j -= 1
            assert j >= 0, f"{name}: no workgroup barrier in front of the ticket"
            j -= 1
            while j >= 0 and not (ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]):
                assert not is_store(ins[j]) and not ins[j].startswith("s_barrier"), \
                    f"{name}: no s_waitcnt vmcnt(0) between the last store and the barrier in front of the ticket ({ins[j]})"
                j -= 1
            assert j >= 0, f"{name}: no s_waitcnt vmcnt(0) in front of the ticket's barrier"
            assert any(is_store(x) and " sc1" in x for x in ins[:j]), f"{name}: no write-through (sc1) payload store ahead of the ticket"
            # ---- consumer side: forwards to the first sc1 load of the last arriver
            k = i + 1
            inv = wait_after_inv = barrier_after = False
            while k < len(ins) and not (ins[k].startswith("global_load") and " sc1" in ins[k]):
                if ins[k].startswith("buffer_inv") and " sc1" in ins[k]:
                    inv = True
                elif inv and ins[k].startswith("s_waitcnt") and "vmcnt(0)" in ins[k]:
                    wait_after_inv = True
                elif wait_after_inv and ins[k].startswith("s_barrier"):
                    barrier_after = True
                elif ins[k].startswith(("global_load", "buffer_load", "flat_load")) and inv:
                    raise AssertionError(f"{name}: a load that is not sc1 follows the acquire before any sc1 load: {ins[k]}")
                k += 1
            assert k < len(ins), f"{name}: no sc1 load of the handed-off partials behind the ticket"
            assert inv and wait_after_inv and barrier_after, f"{name}: acquire / wait / barrier order behind the ticket is wrong"
            ticketed[name] = ticketed.get(name, 0) + 1
    names = " ".join(ticketed)
    assert all(v == 1 and ticketed.get(k, 0) >= 1 for k, v in elections.items()) and len(elections) == 6, elections   # AE TR 32 / 16, Deep-TICA TR 16 / 32 / 64 / 128
    assert "ae_sse_kernel" in names
    for d in (1, 2, 3, 4):
        assert f"tica_stats_rows_kernelILi{d}E" in names
    assert sum(1 for n in ticketed if "gemm_kernel" in n) >= 100   # NT / NN instantiations carry the tail-tile hand-off
    # nothing in the library publishes through a workgroup-scope fence any more
    src = "".join(open(os.path.join(ROOT, "deep_cartograph_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "deep_cartograph_amd", "csrc")) if f.endswith((".h", ".hip")))
    assert not re.search(r'fence\(__ATOMIC_RELEASE,\s*"workgroup"\)', src)
