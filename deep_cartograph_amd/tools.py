"""The three tools of the CV-fit path with the reference's API (argument names, output tree,
CSV formats): train_colvars (tools/train_colvars/train_colvars.py:20-36), traj_projection
(tools/traj_projection/traj_projection.py:19-27), traj_cluster (tools/traj_cluster/traj_cluster.py:
18-28).  Figures, FES estimation and PDB/XTC extraction (matplotlib / MDAnalysis) are outside the
accelerated path and are not produced."""
from __future__ import annotations

import logging
import os
import sys
import time
from pathlib import Path
from typing import Dict, List, Literal, Optional, Union

import numpy as np
import pandas as pd

from . import statistics
from .common import merge_configurations, validate_configuration
from .cv_calculator import CVCalculator, cv_calculators_map
from .schemas import TrainColvarsSchema, TrajClusterSchema, TrajProjectionSchema

logger = logging.getLogger(__name__)


def _as_list(x):
    if x is None:
        return None
    return [x] if isinstance(x, str) else list(x)


def train_colvars(configuration: Dict, train_colvars_paths: Union[str, List[str]], train_topologies: Optional[List[str]] = None,
                  trajectory_names: Optional[List[str]] = None, val_colvars_paths: Optional[Union[str, List[str]]] = None,
                  val_topologies: Optional[List[str]] = None, sup_topologies: Optional[List[str]] = None,
                  sup_traj_names: Optional[List[str]] = None, waypoint_structures: Optional[List[str]] = None,
                  reference_topology: Optional[str] = None, features_list: Optional[List[str]] = None,
                  dimension: Optional[int] = None,
                  cvs: Optional[List[Literal["pca", "ae", "tica", "htica", "deep_tica"]]] = None,
                  frames_per_sample: Optional[int] = 1, output_folder: str = "train_colvars") -> Dict[str, List[str]]:
    """Fit every requested CV, project the training frames, write
    <out>/<cv>/model.zip and <out>/<cv>/traj_data/<traj>/projected_trajectory.csv ('%.4f').
    Returns {cv_name: [csv path per training trajectory]}."""
    t0 = time.time()
    os.makedirs(output_folder, exist_ok=True)
    configuration = validate_configuration(configuration, TrainColvarsSchema, output_folder)
    train_colvars_paths = _as_list(train_colvars_paths)
    val_colvars_paths = _as_list(val_colvars_paths)
    if trajectory_names is None:
        trajectory_names = [Path(p).stem for p in train_colvars_paths]
    cvs_list = list(cvs) if cvs else list(configuration["cvs"])
    unsupported = [c for c in cvs_list if c not in cv_calculators_map]
    for c in unsupported:
        logger.warning(f"CV '{c}' is outside the accelerated path (vae / umap) and is skipped.")
    cvs_list = [c for c in cvs_list if c in cv_calculators_map]
    logger.info(f"Collective variables to compute: {cvs_list}")
    output_paths: Dict[str, List[str]] = {}
    for cv_name in cvs_list:
        cv_folder = os.path.join(output_folder, cv_name)
        # restart support: skip a CV whose model and projections already exist (train_colvars_workflow.py:184-199)
        expected = [os.path.join(cv_folder, "traj_data", n, "projected_trajectory.csv") for n in trajectory_names]
        if os.path.exists(os.path.join(cv_folder, "model.zip")) and all(os.path.exists(p) for p in expected):
            logger.info(f"{cv_name}: already computed, skipping.")
            output_paths[cv_name] = expected
            continue
        merged = merge_configurations(configuration["common"], configuration.get(cv_name, {}))
        calc = cv_calculators_map[cv_name](configuration=merged, output_path=output_folder)
        calc.load_training_data(train_colvars_paths, train_topologies, reference_topology, features_list)
        if val_colvars_paths:
            calc.load_validation_data(val_colvars_paths, val_topologies, reference_topology, features_list)
        df = calc.run(dimension)
        if df is None:
            logger.warning(f"Projected colvars dataframe is empty for {cv_name}. Skipping this CV.")
            continue
        df["traj_label"] = calc.training_data_labels
        paths = []
        # under torch.distributed every rank holds the projection of all frames (run() gathers it): rank 0 writes
        writer = calc.comm.rank == 0
        for i, name in enumerate(trajectory_names):
            traj_folder = os.path.join(cv_folder, "traj_data", name)
            p = os.path.join(traj_folder, "projected_trajectory.csv")
            if writer:
                os.makedirs(traj_folder, exist_ok=True)
                topology = train_topologies[i] if train_topologies else None
                calc.write_plumed_files(topology, os.path.join(traj_folder, "plumed_inputs"), waypoint_structures)
                df_i = df[df["traj_label"] == i].drop("traj_label", axis=1)
                df_i.to_csv(p, index=False, float_format="%.4f")
            paths.append(p)
        calc.comm.barrier()
        output_paths[cv_name] = paths
    logger.info("Elapsed time (Train colvars): %s", time.strftime("%H h %M min %S s", time.gmtime(time.time() - t0)))
    return output_paths


def traj_projection(configuration: Dict, colvars_paths: List[str], topologies: List[str] = None, trajectory_names: List[str] = None,
                    model_paths: List[str] = None, model_traj_paths: Optional[List[List[str]]] = None,
                    output_folder: Optional[str] = "traj_projection") -> Dict[str, List[str]]:
    """Project new colvars files onto saved models: <out>/<cv>/<traj>/projected_trajectory.csv."""
    os.makedirs(output_folder, exist_ok=True)
    validate_configuration(configuration or {}, TrajProjectionSchema, output_folder)
    colvars_paths = _as_list(colvars_paths)
    if trajectory_names is None:
        trajectory_names = [Path(p).stem for p in colvars_paths]
    out: Dict[str, List[str]] = {}
    for model_path in model_paths or []:
        calc = CVCalculator.load(model_path, output_folder)
        paths = []
        for i, (cp, name) in enumerate(zip(colvars_paths, trajectory_names)):
            folder = os.path.join(output_folder, calc.cv_name, name)
            p = os.path.join(folder, "projected_trajectory.csv")
            if os.path.exists(p):  # traj_projection_workflow.py:235-238
                paths.append(p)
                continue
            df = calc.project_colvars([cp], [topologies[i]] if topologies else None)
            if df is None:
                continue
            os.makedirs(folder, exist_ok=True)
            df.to_csv(p, index=False, float_format="%.4f")
            paths.append(p)
        out[calc.cv_name] = paths
    return out


def _read_cv_traj(paths: List[str]) -> pd.DataFrame:
    data = []
    for i, p in enumerate(paths):
        df = pd.read_csv(p)
        df["traj_label"] = i
        data.append(df)
    return pd.concat(data, ignore_index=True)


def traj_cluster(configuration: Dict, cv_traj_paths: Union[str, List[str]], trajectories: Optional[List[str]] = None,
                 topologies: Optional[List[str]] = None, sup_cv_traj_paths: Optional[List[str]] = None,
                 sup_trajectories: Optional[List[str]] = None, sup_topologies: Optional[List[str]] = None,
                 frames_per_sample: Optional[int] = 1, output_folder: str = "traj_cluster") -> Dict[str, List[str]]:
    """Cluster the CV trajectories (CSV in, CSV with cluster / centroid / frame columns out)."""
    os.makedirs(output_folder, exist_ok=True)
    configuration = validate_configuration(configuration or {}, TrajClusterSchema, output_folder)
    if configuration["run"] is False:
        logger.info("traj_cluster workflow set to not run. Exiting...")
        return {}
    cv_traj_paths = _as_list(cv_traj_paths)
    cv_data = _read_cv_traj(cv_traj_paths)
    cv_labels = cv_data.columns[:-1].tolist()
    labels, centroids = statistics.optimize_clustering(cv_data[cv_labels].to_numpy(), configuration)
    cv_data["cluster"] = labels
    cv_data = statistics.find_centroids(cv_data, centroids, cv_labels)
    frames = []
    for i in range(len(cv_traj_paths)):
        n = int((cv_data["traj_label"] == i).sum())
        frames.extend(np.arange(0, n * frames_per_sample, frames_per_sample))
    cv_data["frame"] = frames
    out: Dict[str, List[str]] = {}
    for i in range(len(cv_traj_paths)):
        name = Path(trajectories[i]).stem if trajectories else f"traj_{i}"
        folder = os.path.join(output_folder, name)
        os.makedirs(folder, exist_ok=True)
        p = os.path.join(folder, "projected_trajectory.csv")
        cv_data[cv_data["traj_label"] == i].to_csv(p, index=False)
        out[name] = [p]
    if sup_cv_traj_paths:
        sup = _read_cv_traj(_as_list(sup_cv_traj_paths))
        if sup.shape[1] - 1 != len(cv_labels):
            logger.error("Dimensionality of supplementary collective variable data does not match the original data. Exiting...")
            sys.exit(1)
        sup["cluster"] = statistics.assign_closest_cluster(cv_data[cv_labels].to_numpy(), cv_data["cluster"].to_numpy(),
                                                           sup[cv_labels].to_numpy())
        for i in range(len(sup_cv_traj_paths)):
            name = f"sup_{Path(sup_trajectories[i]).stem}" if sup_trajectories else f"sup_traj_{i}"
            folder = os.path.join(output_folder, name)
            os.makedirs(folder, exist_ok=True)
            p = os.path.join(folder, "projected_trajectory.csv")
            sup[sup["traj_label"] == i].to_csv(p, index=False)
            out[name] = [p]
    return out
