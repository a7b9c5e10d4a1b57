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
* ``schema_defaults.json``  model_dump() of the reference's pydantic schemas for this path.
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
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
