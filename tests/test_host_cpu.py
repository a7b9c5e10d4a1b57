"""CPU tests of the host-side logic: configuration contract, ingestion, small dense solves,
export format and the world-size-2 (gloo) reductions of the frame-sharded path."""
import json
import os
import sys
import zipfile

import numpy as np
import pytest
import torch

from oracle import linear as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- configuration
def test_schema_defaults_match_reference_dump():
    from deep_cartograph_amd.schemas import TrainColvarsSchema, TrajClusterSchema

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "schema_defaults.json")))
    assert json.loads(json.dumps(TrainColvarsSchema().model_dump())) == g["train_colvars"]
    assert json.loads(json.dumps(TrajClusterSchema().model_dump())) == g["traj_cluster"]
    # per-CV override sections are accepted as extra fields
    cfg = TrainColvarsSchema(cvs=["pca"], deep_tica={"lag_time": 7}).model_dump()
    assert cfg["deep_tica"] == {"lag_time": 7}


def test_merge_and_power_of_two(tmp_path):
    from deep_cartograph_amd.common import closest_power_of_two, merge_configurations, validate_configuration
    from deep_cartograph_amd.schemas import TrainColvarsSchema

    common = {"a": 1, "t": {"x": 1, "y": {"z": 2}}}
    out = merge_configurations(common, {"t": {"y": {"z": 3}, "w": 4}, "b": 5})
    assert out == {"a": 1, "t": {"x": 1, "y": {"z": 3}, "w": 4}, "b": 5}
    assert common["t"]["y"]["z"] == 2
    assert [closest_power_of_two(n) for n in (1, 2, 3, 130, 131, 256, 257)] == [1, 1, 2, 128, 128, 128, 256]
    cfg = validate_configuration({"cvs": ["tica"]}, TrainColvarsSchema, str(tmp_path))
    assert cfg["common"]["training"]["general"]["batch_size"] == 32
    assert os.path.exists(tmp_path / "configuration.yml")
    with pytest.raises(SystemExit):
        validate_configuration({"cvs": ["nope"]}, TrainColvarsSchema, None)


# ----------------------------------------------------------------------------- ingestion
def test_colvars_loader_text_and_binary(tmp_path, features):
    from deep_cartograph_amd import colvars

    X, names = features
    p1 = str(tmp_path / "a.dat")
    colvars.write_colvars(p1, X, names)
    p2 = str(tmp_path / "b.npy")
    colvars.write_binary_matrix(p2, X[:50], names)
    A, n1, lab = colvars.load_feature_matrix([p1, p2])
    assert n1 == names and A.shape == (214, 54) and A.dtype == np.float32
    np.testing.assert_allclose(A[:164], X, atol=1e-7)   # text has 8 decimals
    np.testing.assert_array_equal(A[164:], X[:50])
    np.testing.assert_array_equal(lab, [0] * 164 + [1] * 50)
    # start/stop/stride per file, features_list selection and order, 'time' column dropped
    sel = [names[5], names[2]]
    B, n2, _ = colvars.load_feature_matrix(p1, sel, start=3, stop=100, stride=7)
    assert n2 == sel
    np.testing.assert_allclose(B, X[3:100:7][:, [5, 2]], atol=1e-7)
    assert colvars.read_column_names(p1)[0] == "time" and "time" not in colvars.read_column_names(p1, features_only=True)
    with pytest.raises(ValueError):
        colvars.load_feature_matrix(p1, ["not_there"])
    bad = X.copy()
    bad[3, 3] = np.nan
    p3 = str(tmp_path / "c.npy")
    colvars.write_binary_matrix(p3, bad, names)
    with pytest.raises(ValueError):
        colvars.load_feature_matrix(p3)


# ----------------------------------------------------------------------------- dense solves
def test_tica_eigh_and_pca_match_oracle(features):
    from deep_cartograph_amd import linalg

    X, _ = features
    st = ol.feature_stats(X)
    m, r = ol.prepare_normalization(st, "mean_std")
    Xn = ol.normalize(X, m, r).astype(np.float64)
    xt, xl = Xn[:-1], Xn[1:]
    mu = xt.mean(0)
    C0 = (xt - mu).T @ (xt - mu) / len(xt)
    Ct = (xt - mu).T @ (xl - mu) / len(xt)
    C0, Ct = 0.5 * (C0 + C0.T), 0.5 * (Ct + Ct.T)
    ev, V = linalg.tica_eigh(C0, Ct, 1e-6, 2)
    ev_o, V_o = ol.cholesky_eigh(torch.from_numpy(Ct), torch.from_numpy(C0), 1e-6, 2)
    np.testing.assert_allclose(ev, ev_o.numpy(), rtol=1e-8)
    np.testing.assert_allclose(V, V_o.numpy(), atol=1e-8)
    np.testing.assert_allclose(np.linalg.norm(V, axis=0), 1.0, rtol=1e-12)
    assert np.all(V[0] >= 0)
    W = linalg.pca_components(np.cov(Xn.T), 2)
    np.testing.assert_allclose(W, ol.pca_cv(Xn, 2), atol=1e-8)


# ----------------------------------------------------------------------------- export format
def test_torchscript_tree_matches_reference_names(tmp_path, golden_nn):
    from deep_cartograph_amd import export

    g = golden_nn
    lin = [(g[f"deep_tica.param.nn.nn.{i}.weight"], g[f"deep_tica.param.nn.nn.{i}.bias"]) for i in (0, 3, 6)]
    model = export.DeepTICA(export.Normalization(g["deep_tica.buffer.norm_in.mean"], g["deep_tica.buffer.norm_in.range"]),
                            export.FeedForward(lin, ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None]),
                            export.TICA(g["deep_tica.buffer.tica.evecs"], g["deep_tica.buffer.tica.mean"]),
                            export.Normalization(g["deep_tica.buffer.postprocessing.mean"], g["deep_tica.buffer.postprocessing.range"]))
    path = str(tmp_path / "cv_weights.pt")
    export.save_torchscript(model, 54, path)
    m = torch.jit.load(path)
    assert m.original_name == "DeepTICA"
    got_p = {n for n, _ in m.named_parameters()}
    got_b = {n for n, _ in m.named_buffers()}
    assert got_p == {k[len("deep_tica.param."):] for k in g.files if k.startswith("deep_tica.param.")}
    assert got_b == {k[len("deep_tica.buffer."):] for k in g.files if k.startswith("deep_tica.buffer.")}
    assert [c.original_name for c in m.nn.nn.children()] == ["Linear", "LeakyReLU", "Dropout", "Linear", "LeakyReLU", "Dropout", "Linear"]
    X = np.load(os.path.join(ROOT, "tests", "golden", "features_164x54.npz"))["X"]
    with torch.no_grad():
        out = m(torch.from_numpy(np.ascontiguousarray(X))).numpy()
    np.testing.assert_allclose(out, g["deep_tica.output"], atol=1e-6)
    parts = export.read_torchscript(path)
    assert parts["kind"] == "deep_tica" and parts["acts"] == ["leaky_relu", "leaky_relu", None]
    np.testing.assert_array_equal(parts["tica"][1], g["deep_tica.buffer.tica.evecs"])
    # AE tree
    enc = [(g[f"ae.param.encoder.nn.{i}.weight"], g[f"ae.param.encoder.nn.{i}.bias"]) for i in (0, 3, 6)]
    dec = [(g[f"ae.param.decoder.nn.{i}.weight"], g[f"ae.param.decoder.nn.{i}.bias"]) for i in (0, 3, 6)]
    ae = export.AutoEncoderCV(export.Normalization(g["ae.buffer.norm_in.mean"], g["ae.buffer.norm_in.range"]),
                              export.FeedForward(enc, ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None]),
                              export.FeedForward(dec, ["leaky_relu", "leaky_relu", None], [0.0, 0.0, None]),
                              export.Normalization(g["ae.buffer.postprocessing.mean"], g["ae.buffer.postprocessing.range"]))
    path2 = str(tmp_path / "ae.pt")
    export.save_torchscript(ae, 54, path2)
    m2 = torch.jit.load(path2)
    assert {n for n, _ in m2.named_parameters()} == {k[len("ae.param."):] for k in g.files if k.startswith("ae.param.")}
    with torch.no_grad():
        np.testing.assert_allclose(m2(torch.from_numpy(np.ascontiguousarray(X))).numpy(), g["ae.output"], atol=1e-6)


def test_zip_layout(tmp_path):
    from deep_cartograph_amd.common import unzip_files, zip_files

    model = tmp_path / "pca" / "model"
    model.mkdir(parents=True)
    (model / "metadata.json").write_text("{}")
    zip_files(str(tmp_path / "pca" / "model.zip"), str(model))
    with zipfile.ZipFile(tmp_path / "pca" / "model.zip") as z:
        assert z.namelist() == ["model/metadata.json"]   # the reference's model.zip layout
    unzip_files(str(tmp_path / "pca" / "model.zip"), str(tmp_path / "out"))
    assert (tmp_path / "out" / "model" / "metadata.json").exists()


# ----------------------------------------------------------------------------- world size 2 (gloo)
def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from deep_cartograph_amd import hip, parallel

    comm = parallel.Comm()
    rng = np.random.Generator(np.random.PCG64(3))
    n, F, lag = 101, 6, 4
    X = rng.standard_normal((n, F)).astype(np.float32)
    b, e = parallel.shard_bounds(n, world, rank)
    Xl = torch.from_numpy(X[b:e])
    # column statistics: per-shard raw sums (NumPy stand-in for the kernel output) -> global
    X64 = X[b:e].astype(np.float64)
    raw = torch.from_numpy(np.stack([X64.sum(0), (X64 * X64).sum(0), X64.min(0), X64.max(0)]))
    raw = parallel.reduce_col_stats(raw, comm)
    st = hip.finalize_stats(raw, n)
    ref = ol.feature_stats(X)
    ok = np.allclose(st["mean"], ref["mean"], atol=1e-6) and np.allclose(st["std"], ref["std"], rtol=1e-5)
    ok &= np.array_equal(st["min"], ref["min"]) and np.array_equal(st["max"], ref["max"])
    # halo: the union of the per-shard pair sets is the single-process pair set
    Xh, npairs = parallel.append_halo(Xl, lag, comm)
    zt = Xh[:npairs].double().numpy()
    zl = Xh[lag:lag + npairs].double().numpy()
    part = np.concatenate([zt.sum(0), zl.sum(0), (zt.T @ zt).ravel(), (zt.T @ zl).ravel()])
    tot = comm.sum_(torch.from_numpy(part)).numpy()
    P = n - lag
    gt, gl = X[:P].astype(np.float64), X[lag:].astype(np.float64)
    full = np.concatenate([gt.sum(0), gl.sum(0), (gt.T @ gt).ravel(), (gt.T @ gl).ravel()])
    ok &= np.allclose(tot, full, rtol=1e-12, atol=1e-9)
    ok &= int(comm.sum_scalar(npairs)) == P
    _, C0, Ct = hip.covariances_from_raw(tot, P, F)
    mu = gt.mean(0)
    ok &= np.allclose(C0, (gt - mu).T @ (gt - mu) / P, atol=1e-12)
    Ct_ref = (gt - mu).T @ (gl - mu) / P
    ok &= np.allclose(Ct, 0.5 * (Ct_ref + Ct_ref.T), atol=1e-12)
    # min/max and nearest-sample reductions
    mm = parallel.reduce_minmax(torch.from_numpy(np.stack([X64.min(0), X64.max(0)])), comm).numpy()
    ok &= np.array_equal(mm[0], X.min(0).astype(np.float64)) and np.array_equal(mm[1], X.max(0).astype(np.float64))
    C = rng.standard_normal((3, F))
    d = np.linalg.norm(X64[:, None, :] - C[None], axis=2)
    dist_l = torch.from_numpy(d.min(0))
    rows_l = torch.from_numpy(d.argmin(0) + b)
    _, rows = parallel.reduce_nearest(dist_l, rows_l, comm)
    ok &= np.array_equal(rows.numpy(), np.linalg.norm(X.astype(np.float64)[:, None, :] - C[None], axis=2).argmin(0))
    # ragged gather in rank (= frame) order, as the distributed k-means++ seeding uses it
    ok &= np.array_equal(comm.all_gather_rows(Xl).numpy(), X)
    # farthest-point candidates of the empty-cluster relocation: global top-m with payload rows, ties by rank then index
    vals = np.round(np.abs(X64[:, 0]), 1)                     # rounded: ties on purpose
    top = torch.topk(torch.from_numpy(vals), 5)
    gv, gr, gp = parallel.global_topk(top.values, Xl[top.indices].double(), 5, comm)
    all_vals = np.round(np.abs(X.astype(np.float64)[:, 0]), 1)
    ok &= np.array_equal(np.sort(gv.numpy())[::-1], np.sort(all_vals)[::-1][:5]) and bool(np.all(np.diff(gv.numpy()) <= 0))
    ok &= bool(np.all(np.round(np.abs(gp.numpy()[:, 0]), 1) == gv.numpy()))                # payload rows travel with their values
    ok &= bool(np.all((gr.numpy() >= 0) & (gr.numpy() < world)))
    with open(os.path.join(tmpdir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.destroy_process_group()


def test_sharded_reductions_world2(tmp_path):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"ok_{r}").read_text() == "1"


def test_parallel_text_parse_matches_serial(tmp_path):
    """the process-pool COLVAR parser returns exactly the rows of the single pd.read_csv the reference uses,
    in file order, including a comment line in the middle and a missing final newline."""
    from deep_cartograph_amd import colvars

    rng = np.random.default_rng(2)
    X = rng.standard_normal((5003, 7)).astype(np.float32)
    names = ["time"] + [f"d{i}" for i in range(6)]
    path = tmp_path / "c.dat"
    with open(path, "w") as f:
        f.write("#! FIELDS " + " ".join(names) + "\n")
        for i, row in enumerate(X):
            if i == 2500:
                f.write("#! SET something 1\n")
            f.write(" " + " ".join("%.6f" % v for v in row) + ("\n" if i < len(X) - 1 else ""))
    serial = colvars._read_text(str(path), names, workers=1)
    par = colvars._read_text(str(path), names, workers=3, min_bytes=0)
    assert serial.shape == (5003, 7) and par.dtype == np.float32
    np.testing.assert_array_equal(par, serial)
    Xl, kept, _ = colvars.load_feature_matrix(str(path), stride=2)
    np.testing.assert_array_equal(Xl, serial[::2, 1:])
    assert kept == names[1:]


def test_plumed_text_matches_reference_command_module():
    """COMBINE / PRINT / PYTORCH_MODEL lines and the whole linear-CV section against text produced by the reference's
    own modules/plumed/command.py (tests/golden/plumed_text.json, generated by make_golden.py)."""
    import json

    from deep_cartograph_amd.cv_calculator import cv_calculators_map, plumed_combine, plumed_print
    from tests.conftest import GOLDEN, load_golden

    t = json.load(open(os.path.join(GOLDEN, "plumed_text.json")))
    assert plumed_combine("feat_0", ["d1"], [1 / 0.25], [0.1]) == t["combine_basic"]
    assert plumed_combine("pca_1", ["feat_0", "feat_1"], np.array([0.5, -1 / 3])) == t["combine_weights"]
    assert plumed_combine("c", ["a"], periodic=True) == t["combine_periodic"]
    assert plumed_combine("t", ["a", "b"], np.array([0.1, -2.5e-7], dtype=np.float32), np.array([3.0, 1e10], dtype=np.float32)) == t["combine_float32"]
    assert plumed_print(["norm_pca_0", "norm_pca_1"], "pca_out.dat", 1) == t["print"]
    assert plumed_print(["deep_tica.node-0"], "out/colvar.dat", 500, fmt="%.6f") == t["print_stride"]
    # the linear CV section on the arrays of the reference's pca_model.zip
    lin = load_golden("linear_models.npz")
    names = [str(s) for s in load_golden("features_164x54.npz")["names"]]
    calc = cv_calculators_map["pca"]({"dimension": 2, "features_normalization": "mean_std"}, "/tmp/x")
    calc.features_ref_labels = names
    calc.features_norm_mean, calc.features_norm_range = lin["pca.features_norm_mean"], lin["pca.features_norm_range"]
    calc.cv = lin["pca.cv_weights"]
    calc.cv_stats = {"min": np.array(t["linear_cv_stats_min"], dtype=np.float32), "max": np.array(t["linear_cv_stats_max"], dtype=np.float32)}
    assert calc.plumed_cv_lines() == t["linear_cv_section_pca"]
    assert calc.plumed_cv_labels() == ["norm_pca_0", "norm_pca_1"]
    # neural CVs: PYTORCH_MODEL line
    nn = cv_calculators_map["deep_tica"]({"dimension": 2}, "/tmp/x")
    nn.features_ref_labels = names[:5]
    nn.weights_path = "/abs/path/deep_tica_weights.pt"
    assert nn.plumed_cv_lines() == "\n# Collective variable\n" + t["pytorch_model"]


def test_read_torchscript_decomposes_the_reference_model_files():
    """export.read_torchscript on the reference's real deep_tica_model.zip / ae_model.zip (build container only: the files
    never travel): layers, activations and buffers come out as the arrays captured in nn_models.npz."""
    import io
    import tempfile
    import zipfile

    from deep_cartograph_amd import export
    from tests.conftest import load_golden

    base = "/root/reference/deep_cartograph/tests/data/input/models"
    if not os.path.isdir(base):
        pytest.skip("reference not present (GPU box)")
    g = load_golden("nn_models.npz")
    for cv, prefix in (("deep_tica", "nn.nn"), ("ae", "encoder.nn")):
        with zipfile.ZipFile(os.path.join(base, f"{cv}_model.zip")) as z, tempfile.NamedTemporaryFile(suffix=".pt") as f:
            f.write(z.read("model/cv_weights.pt"))
            f.flush()
            parts = export.read_torchscript(f.name)
        assert parts["kind"] == cv and parts["acts"] == ["leaky_relu", "leaky_relu", None]
        for (w, b), i in zip(parts["linears"], (0, 3, 6)):
            np.testing.assert_array_equal(w, g[f"{cv}.param.{prefix}.{i}.weight"])
            np.testing.assert_array_equal(b, g[f"{cv}.param.{prefix}.{i}.bias"])
        np.testing.assert_array_equal(parts["norm_in"][0], g[f"{cv}.buffer.norm_in.mean"])
        np.testing.assert_array_equal(parts["postprocessing"][1], g[f"{cv}.buffer.postprocessing.range"])
        if cv == "deep_tica":
            np.testing.assert_array_equal(parts["tica"][1], g["deep_tica.buffer.tica.evecs"])


def test_calculators_refuse_cpu():
    from deep_cartograph_amd._lib import DcvError
    from deep_cartograph_amd.cv_calculator import cv_calculators_map

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert set(cv_calculators_map) == {"pca", "tica", "htica", "ae", "deep_tica"}
    calc = cv_calculators_map["pca"]({"dimension": 2, "features_normalization": "mean_std"}, "/tmp/x")
    with pytest.raises(DcvError):
        calc.set_training_matrix(np.zeros((10, 4), dtype=np.float32))


def test_optimizer_table_matches_torch_signatures():
    """`optimizer.name` / `kwargs` of the YAML reach `getattr(torch.optim, name)(**kwargs)` in the reference
    (cv_calculator.py:1377-1380).  The calculators' table of optimisers (defaults, accepted keyword arguments) is checked
    against the signatures of the torch.optim classes themselves, and the mapping onto the engine's descriptor is total:
    every optimiser of the table has an engine id, and the two torch.optim classes outside it are refused by name."""
    import inspect

    import torch

    from deep_cartograph_amd import _lib
    from deep_cartograph_amd.cv_calculator import _IMPLEMENTATION_SWITCHES, _OPTIMIZERS, _OPTIMIZERS_REFUSED, NonLinear

    for name, table in _OPTIMIZERS.items():
        sig = inspect.signature(getattr(torch.optim, name).__init__)
        for kw, default in table.items():
            assert kw in sig.parameters, (name, kw)
            d = sig.parameters[kw].default
            if isinstance(default, tuple):
                assert tuple(d) == default, (name, kw, d)
            else:
                assert d == default, (name, kw, d, default)
        extra = set(sig.parameters) - set(table) - set(_IMPLEMENTATION_SWITCHES) - {"self", "params", "maximize"}
        assert not extra, f"{name}: torch accepts {extra} which the table does not know"
        assert name in _lib.OPTIMIZER
        out = NonLinear._engine_optimizer_kwargs(name, dict(table))
        assert out["optimizer"] == name and "lr" in out
    assert NonLinear._engine_optimizer_kwargs("Adam", {**_OPTIMIZERS["Adam"], "decoupled_weight_decay": True, "weight_decay": 0.01})["optimizer"] == "AdamW"
    assert set(_lib.OPTIMIZER) == set(_OPTIMIZERS)
    torch_classes = {n for n in dir(torch.optim) if inspect.isclass(getattr(torch.optim, n)) and issubclass(getattr(torch.optim, n), torch.optim.Optimizer)
                     and n != "Optimizer"}
    assert not (torch_classes - set(_OPTIMIZERS) - set(_OPTIMIZERS_REFUSED)), torch_classes   # every class of the installed torch is either run or refused by name


def test_layer_options_carry_batchnorm():
    """`batchnorm` / `last_layer_batchnorm` (yaml_schemas/train_colvars.py:24-31) become one flag per Linear."""
    from deep_cartograph_amd.cv_calculator import NonLinear

    act, drop, bn = NonLinear._layer_options({"layers": [8, 4], "activation": ["tanh", "relu"], "batchnorm": [True, False], "dropout": [0.1, None],
                                              "last_layer_batchnorm": True, "last_layer_activation": None}, 2)
    assert act == ["tanh", "relu", None] and drop == [0.1, 0.0, 0.0] and bn == [True, False, True]


def test_bench_without_a_gpu_says_so_before_it_spawns_ranks():
    """`bench.py --gpus 2` with no launcher environment on a box without a GPU: the self-launcher counts the devices
    (torch.cuda.device_count() does not initialise anything) and stops with the message -- it does not start ranks
    that would each fail on their own."""
    import subprocess

    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "needs an MI355X" in p.stderr and p.stdout.strip() == ""


def test_bench_sampling_plan():
    """bench.py's roofline sampling: never the first timed step (the one right behind the barrier), the
    two layer-0 products on DIFFERENT steps, five samples of each at the driver's 20 steps, every 8th step in long runs,
    --profile-every 1 = both products of every step but the first."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    plan = bench.Fit.sample_plan(20, 0)
    assert 0 not in plan and len(plan) == 10
    fwd = [i for i, skip in plan.items() if "fwd" not in skip]
    wgr = [i for i, skip in plan.items() if "wgrad" not in skip]
    assert len(fwd) == 5 and len(wgr) == 5 and not set(fwd) & set(wgr)
    long = bench.Fit.sample_plan(2000, 0)
    assert 0 not in long and len(long) == 500 and all(i % 8 in (1, 5) for i in long)
    every = bench.Fit.sample_plan(20, 1)
    assert sorted(every) == list(range(1, 20)) and all(skip == () for skip in every.values())
    assert bench.Fit.sample_plan(1, 0) == {0: ()}          # a one-step run has only that step
    # the timed steps as runs: sampled steps alone, the unsampled ones between them as one dcv_mlp_train_steps call that never
    # crosses an epoch's end (the validation pass follows there)
    for steps, spe in ((20, 976), (400, 195), (7, 3), (2000, 976)):
        picked = bench.Fit.sample_plan(steps, 0)
        runs = list(bench.Fit.runs(steps, picked, spe))
        assert [i for i, _, _ in runs] == [sum(c for _, c, _ in runs[:k]) for k in range(len(runs))] and sum(c for _, c, _ in runs) == steps
        assert all((i % spe) + c <= spe for i, c, _ in runs)
        assert all((c == 1 and i in picked) if sampled else not any(k in picked for k in range(i, i + c)) for i, c, sampled in runs)
    assert max(c for _, c, _ in bench.Fit.runs(400, bench.Fit.sample_plan(400, 0), 195)) == 3


def test_epoch_loop_host_helpers():
    """The host side of an epoch (round 4: it was 70 - 95 % of a fit's wall time): the lazy batch list yields the DictLoader's
    batches, the stacked d x d TICA equals the per-record one bit for bit, torch.randperm does not depend on the thread count
    it is run under."""
    from deep_cartograph_amd import linalg
    from deep_cartograph_amd.cv_calculator import _Batches, _torch_threads

    base = torch.arange(100, 1100)
    b = _Batches("idx", base, 1000, 256)
    assert len(b) == 4 and b.full() == 3 and [b.size(i) for i in range(4)] == [256, 256, 256, 232]
    assert all(kind == "idx" and whole is base for kind, _, whole in b)
    assert torch.equal(torch.cat([v for _, v, _ in b]), base) and torch.equal(b[-1][1], base[768:])
    assert [x[1].numel() for x in b[3:]] == [232] and b[4:] == []
    r = _Batches("range", 7, 513, 256)
    assert list(r) == [("range", 7, 256), ("range", 263, 256), ("range", 519, 1)] and r.full() == 2
    assert len(_Batches("range", 0, 0, 32)) == 0 and _Batches("range", 0, 5, 32).full() == 0
    with pytest.raises(IndexError):
        r[3]
    rng = np.random.default_rng(3)
    for d in (1, 2, 4):
        A = rng.standard_normal((9, d, 3 * d))
        C0 = A @ np.swapaxes(A, -1, -2) / (3 * d)
        B = rng.standard_normal((9, d, d))
        Ct = 0.3 * (B + np.swapaxes(B, -1, -2))
        ev, V = linalg.tica_eigh_stack(C0, Ct, 1e-6)
        for i in range(9):
            e1, v1 = linalg.tica_eigh(C0[i], Ct[i], 1e-6)
            assert np.array_equal(e1, ev[i]) and np.array_equal(v1, V[i])
    assert linalg.tica_eigh_stack(np.zeros((0, 2, 2)), np.zeros((0, 2, 2)))[0].shape == (0, 2)
    before = torch.get_num_threads()
    torch.manual_seed(11)
    a = torch.randperm(50000)
    torch.manual_seed(11)
    with _torch_threads(1):
        assert torch.get_num_threads() == 1
        c = torch.randperm(50000)
    assert torch.get_num_threads() == before and torch.equal(a, c)
