"""Generate the golden fixtures under tests/golden/ from the reference's OWN test data.

Run in the build container only (it reads /root/reference, which never travels):

    python tests/golden/make_golden.py

What is stored is *data* (inputs and expected outputs), never reference source:

* ``features_164x54.npz``   the 164x54 float32 feature matrix the reference's tests train on
  (tests/data/reference/compute_features/virtual_dihedrals.dat restricted/ordered by
  tests/data/reference/filter_features/filtered_virtual_dihedrals.txt) and its column names;
* ``train_colvars_golden.npz``  the six reference/train_colvars/*_projected_trajectory.csv;
* ``linear_models.npz``     arrays inside input/models/{pca,tica,htica}_model.zip;
* ``nn_models.npz``         parameters, buffers and outputs of the TorchScript files inside
  input/models/{deep_tica,ae}_model.zip, evaluated here with torch.jit.load;
* ``cluster_golden.npz``    reference/traj_cluster/*.csv (labels, centroid flags) plus the
  outputs of the reference's importable ``statistics`` module (hierarchical as in its tests,
  and k-means, which the reference's tests do not pin) on those CSVs and on seeded
  synthetic sets;
* ``schema_defaults.json``  model_dump() of the reference's pydantic schemas for this path;
* ``plumed_text.json``      PLUMED input text produced by the reference's own ``modules/plumed/command.py`` (loaded
  standalone with importlib: it needs only the standard library and NumPy) for the CV section the assembler writes
  (``plumed/input/assembler.py:333-431``: the comment lines are its literals) on the arrays of ``pca_model.zip`` and for
  a PYTORCH_MODEL CV, plus single COMBINE / PRINT lines.
"""
import io
import json
import os
import sys
import zipfile

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
DATA = os.path.join(REF, "deep_cartograph", "tests", "data")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

CVS = ["pca", "tica", "htica", "ae", "deep_tica", "vae"]


def read_colvars(path):
    with open(path) as f:
        names = f.readline().split()[2:]
    return pd.read_csv(path, sep=r"\s+", dtype=np.float32, comment="#", header=None, names=names)


def main():
    # ------------------------------------------------------------------ features
    df = read_colvars(os.path.join(DATA, "reference", "compute_features", "virtual_dihedrals.dat"))
    with open(os.path.join(DATA, "reference", "filter_features", "filtered_virtual_dihedrals.txt")) as f:
        feats = f.read().split()
    X = np.ascontiguousarray(df[feats].to_numpy(dtype=np.float32))
    assert X.shape == (164, 54), X.shape
    np.savez_compressed(os.path.join(OUT, "features_164x54.npz"), X=X, names=np.array(feats))

    # ------------------------------------------------------------------ projections
    proj = {}
    for cv in CVS:
        proj[cv] = pd.read_csv(os.path.join(DATA, "reference", "train_colvars", f"{cv}_projected_trajectory.csv")).to_numpy(dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "train_colvars_golden.npz"), **proj)

    # ------------------------------------------------------------------ linear models
    lin = {}
    for cv in ["pca", "tica", "htica"]:
        with zipfile.ZipFile(os.path.join(DATA, "input", "models", f"{cv}_model.zip")) as z:
            for name in ["cv_weights", "cv_norm_mean", "cv_norm_range", "features_norm_mean", "features_norm_range"]:
                lin[f"{cv}.{name}"] = np.load(io.BytesIO(z.read(f"model/{name}.npy")))
            lin[f"{cv}.metadata"] = np.array(z.read("model/metadata.json").decode())
    np.savez_compressed(os.path.join(OUT, "linear_models.npz"), **lin)

    # ------------------------------------------------------------------ NN models
    nn = {}
    Xt = torch.from_numpy(X)
    for cv in ["deep_tica", "ae"]:
        with zipfile.ZipFile(os.path.join(DATA, "input", "models", f"{cv}_model.zip")) as z:
            m = torch.jit.load(io.BytesIO(z.read("model/cv_weights.pt")))
        m.eval()
        for n, p in m.named_parameters():
            nn[f"{cv}.param.{n}"] = p.detach().numpy()
        for n, b in m.named_buffers():
            nn[f"{cv}.buffer.{n}"] = b.detach().numpy()
        with torch.no_grad():
            nn[f"{cv}.output"] = m(Xt).numpy()
    np.savez_compressed(os.path.join(OUT, "nn_models.npz"), **nn)

    # ------------------------------------------------------------------ clustering
    from deep_cartograph.modules.statistics import statistics as ref_stats
    from deep_cartograph.yaml_schemas.traj_cluster import TrajClusterSchema
    from deep_cartograph.yaml_schemas.train_colvars import TrainColvarsSchema

    cl = {}
    for cv in CVS:
        g = pd.read_csv(os.path.join(DATA, "reference", "traj_cluster", f"{cv}_projected_trajectory.csv"))
        cl[f"{cv}.golden_cluster"] = g["cluster"].to_numpy(dtype=np.int64)
        cl[f"{cv}.golden_centroid"] = g["centroid"].to_numpy(dtype=bool)
        P = proj[cv]
        # (1) the reference's test configuration: defaults => hierarchical / complete / k in [3, 10]
        settings = TrajClusterSchema().model_dump()
        labels, cents = ref_stats.optimize_clustering(P.copy(), settings)
        cl[f"{cv}.hier_labels"] = labels.astype(np.int64)
        cl[f"{cv}.hier_centroids"] = cents
        dfp = pd.DataFrame(P.copy(), columns=["a", "b"])
        cl[f"{cv}.hier_centroid_flag"] = ref_stats.find_centroids(dfp, cents, ["a", "b"])["centroid"].to_numpy(dtype=bool)
        # (2) k-means through the reference module (unpinned by the reference's own tests)
        settings = TrajClusterSchema(algorithm="kmeans").model_dump()
        labels, cents = ref_stats.optimize_clustering(P.copy(), settings)
        cl[f"{cv}.kmeans_opt_labels"] = labels.astype(np.int64)
        cl[f"{cv}.kmeans_opt_centroids"] = cents
        for k in (3, 6):
            s = {"algorithm": "kmeans", "num_clusters": k, "n_init": 5}
            labels, cents = ref_stats.cluster_data(P.copy(), s)
            cl[f"{cv}.kmeans_k{k}_labels"] = labels.astype(np.int64)
            cl[f"{cv}.kmeans_k{k}_centroids"] = cents
        init = P[[0, 40, 80, 120]].copy()
        labels, cents = ref_stats.cluster_data(P.copy(), {"algorithm": "kmeans"}, initial_centroids=init)
        cl[f"{cv}.kmeans_init_labels"] = labels.astype(np.int64)
        cl[f"{cv}.kmeans_init_centroids"] = cents

    # seeded synthetic mixtures (values with 4 decimals, as after the CSV seam)
    for tag, (n, d, k, seed) in {"syn_a": (20000, 4, 6, 7), "syn_b": (5000, 2, 4, 11), "syn_c": (3000, 3, 8, 5)}.items():
        rng = np.random.Generator(np.random.PCG64(seed))
        mu = rng.uniform(-0.8, 0.8, size=(k, d))
        comp = rng.integers(0, k, size=n)
        P = np.clip(mu[comp] + 0.08 * rng.standard_normal((n, d)), -1, 1)
        P = np.round(P, 4)
        cl[f"{tag}.points"] = P
        init = P[rng.choice(n, size=k, replace=False)].copy()
        cl[f"{tag}.init"] = init
        labels, cents = ref_stats.cluster_data(P.copy(), {"algorithm": "kmeans"}, initial_centroids=init.copy())
        cl[f"{tag}.init_labels"] = labels.astype(np.int64)
        cl[f"{tag}.init_centroids"] = cents
        labels, cents = ref_stats.cluster_data(P.copy(), {"algorithm": "kmeans", "num_clusters": k, "n_init": 3})
        cl[f"{tag}.pp_labels"] = labels.astype(np.int64)
        cl[f"{tag}.pp_centroids"] = cents
        dfp = pd.DataFrame(P.copy(), columns=[f"c{i}" for i in range(d)])
        cl[f"{tag}.pp_centroid_flag_rows"] = np.where(
            ref_stats.find_centroids(dfp, cents, list(dfp.columns))["centroid"].to_numpy(dtype=bool))[0]
    np.savez_compressed(os.path.join(OUT, "cluster_golden.npz"), **cl)

    # ------------------------------------------------------------------ schema defaults
    with open(os.path.join(OUT, "schema_defaults.json"), "w") as f:
        json.dump({"train_colvars": TrainColvarsSchema().model_dump(),
                   "traj_cluster": TrajClusterSchema().model_dump()}, f, indent=1, sort_keys=True)
    # ------------------------------------------------------------------ PLUMED text
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_plumed_command", os.path.join(REF, "deep_cartograph", "modules", "plumed", "command.py"))
    cmd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmd)
    text = {}
    text["combine_basic"] = cmd.combine("feat_0", ["d1"], [1 / 0.25], [0.1])
    text["combine_weights"] = cmd.combine("pca_1", ["feat_0", "feat_1"], np.array([0.5, -1 / 3]))
    text["combine_periodic"] = cmd.combine("c", ["a"], periodic=True)
    text["combine_float32"] = cmd.combine("t", ["a", "b"], np.array([0.1, -2.5e-7], dtype=np.float32), np.array([3.0, 1e10], dtype=np.float32))
    text["print"] = cmd.print(["norm_pca_0", "norm_pca_1"], "pca_out.dat", 1)
    text["print_stride"] = cmd.print(["deep_tica.node-0"], "out/colvar.dat", 500, fmt="%.6f")
    text["pytorch_model"] = cmd.pytorch_model("deep_tica", feats[:5], "/abs/path/deep_tica_weights.pt")
    # the linear CV section as assembler.add_linear_cv composes it, on the arrays of pca_model.zip; cv_stats in float32 as
    # LinearCalculator.normalize_cv leaves them (min = mean - range, max = mean + range of the stored normalisation)
    fm, fr = lin["pca.features_norm_mean"], lin["pca.features_norm_range"]
    W = lin["pca.cv_weights"]
    cmin = (lin["pca.cv_norm_mean"] - lin["pca.cv_norm_range"]).astype(np.float32)
    cmax = (lin["pca.cv_norm_mean"] + lin["pca.cv_norm_range"]).astype(np.float32)
    sec = "\n# Normalized features\n"
    labels = []
    for i, f_ in enumerate(feats):
        sec += cmd.combine(f"feat_{i}", [f_], [1 / fr[i]], [fm[i]])
        labels.append(f"feat_{i}")
    sec += "\n# Collective variable\n"
    for i in range(W.shape[1]):
        sec += cmd.combine(f"pca_{i}", labels, W[:, i])
    off, sc = (cmin + cmax) / 2, 2 / (cmax - cmin)
    sec += "\n# Normalized Collective variable\n"
    for i in range(W.shape[1]):
        sec += cmd.combine(f"norm_pca_{i}", [f"pca_{i}"], [sc[i]], [off[i]])
    text["linear_cv_section_pca"] = sec
    text["linear_cv_stats_min"] = [float(v) for v in cmin]
    text["linear_cv_stats_max"] = [float(v) for v in cmax]
    with open(os.path.join(OUT, "plumed_text.json"), "w") as f:
        json.dump(text, f, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
