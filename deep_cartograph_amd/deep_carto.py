"""`deep_carto`-style entry point for the accelerated part of the workflow:
train_colvars -> (traj_projection of supplementary data) -> traj_cluster, starting from
pre-computed feature matrices (PLUMED COLVAR text or the binary .npy fast path).  The
reference's steps 0-3 (geometry analysis, augmentation, PLUMED featurisation, feature filtering;
deep_carto.py:191-305) need MDAnalysis and the plumed binary and are out of scope, so the YAML
keeps the reference's `train_colvars` / `traj_cluster` sections and the CLI takes colvars
files where the reference takes trajectories.

    python -m deep_cartograph_amd.deep_carto -conf config.yml -colvars a.dat b.dat -out run1 \
           [-dim 2] [-cvs pca tica deep_tica] [-sup_colvars c.dat] [-features feats.txt] [-restart]
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time
from typing import Dict, List, Optional

from .common import get_unique_path, read_configuration, read_features_list
from .tools import traj_cluster, traj_projection, train_colvars

logger = logging.getLogger("deep_cartograph")


def deep_cartograph(configuration: Dict, colvars_paths: List[str], sup_colvars_paths: Optional[List[str]] = None,
                    features_list: Optional[List[str]] = None, dimension: Optional[int] = None, cvs: Optional[List[str]] = None,
                    restart: bool = False, output_folder: Optional[str] = None) -> Dict[str, Dict]:
    """train_colvars -> traj_projection (supplementary colvars) -> traj_cluster per CV
    (reference deep_carto.py:307-361).  `restart` reuses the output folder and skips what exists."""
    t0 = time.time()
    output_folder = output_folder or "deep_cartograph"
    if not restart:
        output_folder = get_unique_path(output_folder)
    os.makedirs(output_folder, exist_ok=True)
    tc_out = os.path.join(output_folder, "train_colvars")
    cv_paths = train_colvars(configuration=configuration.get("train_colvars", {}), train_colvars_paths=colvars_paths,
                             features_list=features_list, dimension=dimension, cvs=cvs, output_folder=tc_out)
    sup_paths: Dict[str, List[str]] = {}
    if sup_colvars_paths:
        models = [os.path.join(tc_out, cv, "model.zip") for cv in cv_paths]
        sup_paths = traj_projection(configuration=configuration.get("traj_projection", {}), colvars_paths=sup_colvars_paths,
                                    model_paths=models, output_folder=os.path.join(output_folder, "traj_projection"))
    clusters = {}
    for cv, paths in cv_paths.items():
        clusters[cv] = traj_cluster(configuration=configuration.get("traj_cluster", {}), cv_traj_paths=paths,
                                    sup_cv_traj_paths=sup_paths.get(cv), output_folder=os.path.join(output_folder, "traj_cluster", cv))
    logger.info("Total elapsed time: %s", time.strftime("%H h %M min %S s", time.gmtime(time.time() - t0)))
    return {"train_colvars": cv_paths, "traj_projection": sup_paths, "traj_cluster": clusters}


def main(argv=None):
    p = argparse.ArgumentParser("deep_carto (MI355X CV-fit path)")
    p.add_argument("-conf", "-configuration", dest="configuration_path", required=True, help="YAML configuration")
    p.add_argument("-colvars", dest="colvars", nargs="+", required=True, help="training feature matrices (COLVAR text or .npy)")
    p.add_argument("-sup_colvars", dest="sup_colvars", nargs="*", default=None, help="supplementary feature matrices to project")
    p.add_argument("-features", dest="features_path", default=None, help="file with the feature names to use (one per line)")
    p.add_argument("-dim", "-dimension", dest="dimension", type=int, default=None)
    p.add_argument("-cvs", nargs="+", default=None)
    p.add_argument("-restart", action="store_true")
    p.add_argument("-out", "-output", dest="output_folder", default=None)
    p.add_argument("-v", "-verbose", dest="verbose", action="store_true")
    a = p.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if a.verbose else logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    cfg = read_configuration(a.configuration_path)
    deep_cartograph(cfg, a.colvars, a.sup_colvars, read_features_list(a.features_path), a.dimension, a.cvs, a.restart, a.output_folder)


if __name__ == "__main__":
    main(sys.argv[1:])
