"""Collective-variable calculators: host-side mirror of the reference's
deep_cartograph/modules/cv_learning/cv_calculator.py (class tree, method names, argument meaning,
output files and error behaviour) over the HIP kernels of libdcv.so.

    cv_calculators_map[name](configuration, output_path)
        .load_training_data(paths, topologies, ref_topology, features_list)
        .run(dimension) -> DataFrame of the projected training frames (None = fit failed)
        .project_colvars(...) / .project_data(...)
    CVCalculator.load(model.zip, output_path)

The frames x features matrix lives on the GPU from load_training_data on; statistics,
normalisation, covariance, projection and the AE / Deep-TICA training are HIP kernels, the
F x F (or d x d) eigen-solves run on the host in float64 (linalg.py).  With torch.distributed
initialised every rank holds a contiguous block of frames and the small result buffers are
all-reduced (parallel.py).  There is no CPU fallback.
"""
from __future__ import annotations

import copy
import json
import logging
import os
import shutil
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import pandas as pd
import torch

from . import export, hip, linalg
from .colvars import load_feature_matrix
from .common import closest_power_of_two, remove_files, unzip_files, zip_files
from .parallel import Comm, append_halo, reduce_col_stats, reduce_minmax

logger = logging.getLogger(__name__)


def _to_host(t: torch.Tensor) -> torch.Tensor:
    """Device -> host copy of a projection through page-locked memory of torch's caching host allocator: a fresh
    pageable buffer of tens of MB costs its page faults on every call (measured 3.5 ms against 60-90 ms for 5M x 2)."""
    host = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
    host.copy_(t)
    return host


def _dev(a, device, dtype=torch.float32) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a), dtype=dtype).to(device)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise hip.DcvError("deep_cartograph_amd needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class CVCalculator(ABC):
    """Base class (reference: cv_calculator.py:23-747)."""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        self.configuration: Dict = copy.deepcopy(configuration) if configuration is not None else {}
        self.architecture_config: Dict = self.configuration.get("architecture", {})
        self.training_reading_settings: Dict = self.configuration.get("input_colvars", {}) or {}
        self.feats_norm_mode = self.configuration.get("features_normalization", None)
        self.bias: Dict = self.configuration.get("bias", {})
        self.ref_topology_path: Optional[str] = None
        self.training_data: Optional[torch.Tensor] = None      # device, [frames of this rank, F]
        self.training_data_labels: Optional[np.ndarray] = None
        self.validation_data: Optional[torch.Tensor] = None
        self.projection_data_labels: Optional[np.ndarray] = None
        self.features_ref_labels: List[str] = []
        self.features_stats: Dict[str, np.ndarray] = {}
        self.features_norm_mean: Optional[np.ndarray] = None
        self.features_norm_range: Optional[np.ndarray] = None
        self.num_features: int = 0
        self.num_frames_global: int = 0
        self.cv = None
        self.cv_dimension: int = self.configuration.get("dimension")
        self.cv_labels: List[str] = []
        self.cv_name: str = None
        self.cv_range: List[Tuple[float, float]] = []
        self.parent_output_path: str = output_path
        self.plumed_files: List[str] = []
        self.temp_model_path: Optional[str] = None
        self.comm = Comm()

    def __del__(self):
        try:
            if self.temp_model_path and os.path.exists(self.temp_model_path):
                shutil.rmtree(self.temp_model_path)
        except Exception:
            pass

    # ------------------------------------------------------------------ loading a saved model
    @classmethod
    def load(cls, model_path: str, output_path: str):
        """Factory: model.zip -> the right calculator, ready to project (reference :92-149)."""
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"Model file not found: {model_path}")
        temp_model_path = os.path.join(output_path, "model")
        unzip_files(model_path, output_path)
        metadata_path = os.path.join(temp_model_path, "metadata.json")
        cv_name = None
        if os.path.exists(metadata_path):
            with open(metadata_path) as f:
                cv_name = json.load(f).get("cv_name")
        else:
            logger.error(f"Metadata file not found in the model: {metadata_path}")
        if not cv_name:
            raise ValueError("Could not determine the CV name from the model file.")
        klass = cv_calculators_map.get(cv_name)
        if not klass:
            raise TypeError(f"Unknown CV calculator name: {cv_name}")
        inst = klass(output_path=output_path)
        inst._load_from_folder(temp_model_path)
        inst.temp_model_path = temp_model_path
        return inst

    def _load_from_folder(self, folder_path: str):
        with open(os.path.join(folder_path, "metadata.json")) as f:
            meta = json.load(f)
        self.cv_dimension = meta.get("cv_dimension")
        self.cv_name = meta.get("cv_name")
        self.set_labels()
        self.model_output_folder = os.path.join(self.parent_output_path, self.cv_name, "model")
        if os.path.exists(self.model_output_folder):
            shutil.rmtree(self.model_output_folder)
        shutil.copytree(folder_path, self.model_output_folder)
        with open(os.path.join(self.model_output_folder, "features_labels.txt")) as f:
            self.features_ref_labels = f.read().strip().split("\n")
        self.num_features = len(self.features_ref_labels)
        ref = os.path.join(self.model_output_folder, "ref_topology.pdb")
        self.ref_topology_path = ref if os.path.exists(ref) else None

    def create_output_folders(self):
        self.output_path = Path(self.parent_output_path) / self.cv_name
        self.sensitivity_output_folder = self.output_path / "sensitivity_analysis"
        self.training_output_folder = self.output_path / "training"
        self.model_output_folder = self.output_path / "model"
        if self.comm.rank == 0:
            for p in (self.output_path, self.sensitivity_output_folder, self.training_output_folder, self.model_output_folder):
                p.mkdir(parents=True, exist_ok=True)
        self.comm.barrier()

    # ------------------------------------------------------------------ data
    def _read(self, paths, features_list):
        """Colvars files -> (this rank's contiguous block of the frames, names, labels of ALL frames)."""
        s = self.training_reading_settings
        shard = (self.comm.world, self.comm.rank) if self.comm.active else None
        return load_feature_matrix(paths, features_list, start=s.get("start", 0), stop=s.get("stop"), stride=s.get("stride", 1), shard=shard)

    def load_training_data(self, train_colvars_paths: List[str], train_topology_paths: Optional[List[str]] = None,
                           ref_topology_path: Optional[str] = None, features_list: Optional[List[str]] = None):
        """Read the colvars files, keep the matrix on the GPU, compute the feature statistics
        with the HIP column-statistics kernel (reference :248-300)."""
        self.ref_topology_path = ref_topology_path
        if train_topology_paths is not None and self.ref_topology_path is None:
            self.ref_topology_path = train_topology_paths[0]
        logger.info("Reading training data from colvars files...")
        X, names, labels = self._read(train_colvars_paths, features_list)
        self.set_training_matrix(X, names, labels)

    def set_training_matrix(self, X, feature_names: Optional[List[str]] = None, labels: Optional[np.ndarray] = None):
        """Entry point for callers that already hold the (shard of the) feature matrix: a float32
        array / tensor of this rank's frames (under torch.distributed: the rank's contiguous block, blocks
        in rank order forming the trajectory).  A device tensor is adopted without a copy.  `labels`:
        trajectory index of every frame of ALL ranks."""
        dev = _device()
        if isinstance(X, torch.Tensor):
            Xd = X.to(device=dev, dtype=torch.float32)
        else:
            Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(dev)
        if Xd.dim() != 2:
            raise ValueError("feature matrix must be 2-D")
        self.training_data = Xd.contiguous()
        n_local, F = self.training_data.shape
        self.features_ref_labels = list(feature_names) if feature_names is not None else [f"f{i}" for i in range(F)]
        self.num_features = F
        logger.info(f"Number of features: {self.num_features}")
        raw = reduce_col_stats(hip.col_stats_raw(self.training_data), self.comm)
        self.num_frames_global = int(round(self.comm.sum_scalar(n_local, device=dev)))
        self.training_data_labels = labels if labels is not None else np.zeros(self.num_frames_global, dtype=np.int64)
        if len(self.training_data_labels) != self.num_frames_global:
            raise ValueError(f"{len(self.training_data_labels)} trajectory labels for {self.num_frames_global} frames")
        self.features_stats = hip.finalize_stats(raw, self.num_frames_global)
        self.features_norm_mean, self.features_norm_range = self.prepare_normalization()

    def load_validation_data(self, val_colvars_paths: List[str], val_topology_paths: Optional[List[str]] = None,
                             ref_topology_path: Optional[str] = None, features_list: Optional[List[str]] = None):
        """Separate validation set (reference :208-246): the fit then trains on ALL training samples and
        evaluates on these frames instead of splitting (:1485-1492)."""
        logger.info("Reading validation data from colvars files...")
        X, _, _ = self._read(val_colvars_paths, features_list)
        self.set_validation_matrix(X)

    def set_validation_matrix(self, X):
        dev = _device()
        Xd = X.to(device=dev, dtype=torch.float32) if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(dev)
        if Xd.dim() != 2 or (self.num_features and Xd.shape[1] != self.num_features):
            raise ValueError("validation matrix must be 2-D with the training features")
        self.validation_data = Xd.contiguous()

    def cv_ready(self) -> bool:
        return self.cv is not None

    def prepare_normalization(self) -> Tuple[np.ndarray, np.ndarray]:
        """(mean, range) of the chosen mode; |range| < 1e-8 -> 1 (reference :308-363)."""
        st = self.features_stats
        mode = self.feats_norm_mode
        if mode is None:
            means = np.zeros(len(st["mean"]))
            ranges = np.ones(len(st["mean"]))
        elif mode == "mean_std":
            means, ranges = st["mean"], st["std"]
        elif mode == "min_max_range1":
            means, ranges = st["min"], st["max"] - st["min"]
        elif mode == "min_max_range2":
            means, ranges = (st["min"] + st["max"]) / 2, (st["max"] - st["min"]) / 2
        else:
            logger.error(f"Normalization mode {mode} not recognized. Exiting...")
            raise ValueError(f"Normalization mode {mode} not recognized.")
        ranges = np.array(ranges, copy=True)
        for i in np.where(np.abs(ranges) < 1e-8)[0]:
            ranges[i] = 1.0
            logger.warning(f"Range for feature {i} is close to zero. Setting it to 1.0.")
        return np.asarray(means), ranges

    # ------------------------------------------------------------------ driver
    def run(self, cv_dimension: Union[int, None] = None) -> Union[pd.DataFrame, None]:
        """compute_cv -> normalize_cv -> project the training frames -> save_model ->
        sensitivity_analysis (reference :366-414).  Returns None when the fit failed."""
        if self.training_data is None:
            logger.error("Training data not loaded. Cannot compute CV.")
            return None
        self.create_output_folders()
        if cv_dimension:
            self.cv_dimension = cv_dimension
        self.compute_cv()
        self.set_labels()
        projection_df = None
        if self.cv is not None:
            self.normalize_cv()
            projection = self.project_data(self.training_data, normalize_data=False)
            if self.comm.active:   # every rank returns the projection of ALL training frames (rank order = frame order)
                projection = _to_host(self.comm.all_gather_rows(projection.to(self.training_data.device)))
            if self.comm.rank == 0:
                self.save_model()
            self.sensitivity_analysis()   # every rank computes (collectives inside), rank 0 writes
            self.comm.barrier()
            projection_df = pd.DataFrame(projection.numpy(), columns=self.cv_labels)
        return projection_df

    @abstractmethod
    def compute_cv(self):
        raise NotImplementedError

    @abstractmethod
    def save_weights(self, weights_path: str):
        raise NotImplementedError

    def save_model(self):
        """Files common to every calculator: metadata.json, features_labels.txt,
        ref_topology.pdb (reference :436-452)."""
        with open(os.path.join(self.model_output_folder, "metadata.json"), "w") as f:
            json.dump({"cv_name": self.cv_name, "cv_dimension": self.cv_dimension}, f)
        np.savetxt(os.path.join(self.model_output_folder, "features_labels.txt"), self.features_ref_labels, fmt="%s")
        if self.ref_topology_path is not None and os.path.exists(self.ref_topology_path):
            shutil.copyfile(self.ref_topology_path, os.path.join(self.model_output_folder, "ref_topology.pdb"))

    @abstractmethod
    def get_cv_parameters(self) -> Dict:
        raise NotImplementedError

    @abstractmethod
    def get_cv_type(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def project_data(self, data, normalize_data: bool = True) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def normalize_cv(self):
        raise NotImplementedError

    def sensitivity_analysis(self):
        return

    def project_colvars(self, colvars_paths: Union[List[str], str], topology_paths: Union[List[str], str, None] = None
                        ) -> Union[pd.DataFrame, None]:
        """Project colvars files with a fitted / loaded CV (reference :478-526).  Unlike the
        reference a model without reference topology still projects:
        the files must then carry the training feature names."""
        if self.ref_topology_path is None:
            logger.warning("Reference topology not set. Make sure the colvars file matches the training data.")
        X, _, labels = load_feature_matrix(colvars_paths, self.features_ref_labels)
        self.projection_data_labels = labels
        projected = self.project_data(torch.from_numpy(X))
        return pd.DataFrame(projected.numpy(), columns=self.cv_labels)

    def set_labels(self):
        self.cv_labels = [f"{cv_components_map[self.cv_name]} {i + 1}" for i in range(self.cv_dimension)]

    def get_labels(self) -> List[str]:
        return self.cv_labels

    def get_cv_dimension(self) -> int:
        return self.cv_dimension

    def get_range(self) -> List[Tuple[float, float]]:
        return self.cv_range

    def write_plumed_files(self, topology: Optional[str], output_folder: str, waypoint_structures=None):
        """PLUMED input that tracks this CV (reference cv_calculator.py:545-681 -> ComputeCVBuilder).
        Without a topology the reference returns early (:569-571).  Written here: the header, the CV
        section exactly as the reference's assembler composes it (assembler.py:333-431 with
        command.combine / command.pytorch_model, command.py:357-420, 1149-1178) and the PRINT line
        (command.py:520-564, traj_stride 1), zipped as plumed_<cv>_unbiased.zip together with the
        TorchScript weights of a neural CV.  The MOLINFO / WHOLEMOLECULES / per-feature commands and
        the feature-name translation between topologies need MDAnalysis (out of scope): the feature
        labels are used as the PLUMED argument names, and the enhanced-sampling input is not built."""
        if topology is None:
            logger.warning("Topology not provided. Skipping PLUMED files creation.")
            return
        os.makedirs(output_folder, exist_ok=True)
        self.plumed_files = []
        if self.get_cv_type() == "non-linear":
            self.weights_path = os.path.join(output_folder, f"{self.cv_name}_weights.pt")
            self.save_weights(self.weights_path)
            self.plumed_files.append(self.weights_path)
        path = os.path.join(output_folder, f"plumed_input_{self.cv_name}.dat")
        text = "# PLUMED input file generated with Deep Cartograph\n"
        text += self.plumed_cv_lines()
        text += "\n" + plumed_print(self.plumed_cv_labels(), f"{self.cv_name}_out.dat", 1)
        with open(path, "w") as f:
            f.write(text)
        self.plumed_files.append(path)
        zip_files(os.path.join(output_folder, f"plumed_{self.cv_name}_unbiased.zip"), *self.plumed_files)
        remove_files(*self.plumed_files)

    def plumed_cv_lines(self) -> str:
        raise NotImplementedError

    def plumed_cv_labels(self) -> List[str]:
        raise NotImplementedError


def plumed_combine(label: str, arguments, coefficients=None, parameters=None, periodic: bool = False) -> str:
    """PLUMED COMBINE line in the reference's format (command.py:357-420): %.17g coefficients / parameters."""
    line = label + ": COMBINE ARG=" + ",".join(arguments)
    if coefficients is not None:
        line += " COEFFICIENTS=" + ",".join(f"{c:.17g}" for c in coefficients)
    if parameters is not None:
        line += " PARAMETERS=" + ",".join(f"{a:.17g}" for a in parameters)
    return line + (" PERIODIC=YES" if periodic else " PERIODIC=NO") + "\n"


def plumed_print(arguments, file_path: str, stride: int = 1, fmt: str = "%.4f") -> str:
    """PLUMED PRINT line in the reference's format (command.py:520-564)."""
    return "PRINT ARG=" + ",".join(arguments) + " FILE=" + file_path + " STRIDE=" + str(stride) + f" FMT={fmt}\n"


# ======================================================================================= linear
class LinearCalculator(CVCalculator):
    """Linear CVs: weights F x d (reference :749-1047)."""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv: Optional[np.ndarray] = None
        self.weights_path: Optional[str] = None
        self.cv_stats: Dict[str, np.ndarray] = {}
        self.cv_norm_mean: Optional[np.ndarray] = None
        self.cv_norm_range: Optional[np.ndarray] = None

    def _load_from_folder(self, folder_path: str):
        super()._load_from_folder(folder_path)
        f = self.model_output_folder
        self.cv = np.load(os.path.join(f, "cv_weights.npy"))
        self.cv_norm_mean = np.load(os.path.join(f, "cv_norm_mean.npy"))
        self.cv_norm_range = np.load(os.path.join(f, "cv_norm_range.npy"))
        self.features_norm_mean = np.load(os.path.join(f, "features_norm_mean.npy"))
        self.features_norm_range = np.load(os.path.join(f, "features_norm_range.npy"))

    def set_training_matrix(self, X, feature_names=None, labels=None):
        super().set_training_matrix(X, feature_names, labels)
        # linear models normalise the stored training matrix in place (reference :800-804)
        dev = self.training_data.device
        hip.normalize(self.training_data, _dev(self.features_norm_mean, dev), _dev(self.features_norm_range, dev),
                      out=self.training_data)

    def normalize_data(self, data: torch.Tensor, normalizing_mean, normalizing_range) -> torch.Tensor:
        """(data - mean) / range in float32 on the GPU (in place, as the reference)."""
        dev = _device()
        d = data if data.is_cuda else data.to(dev)
        return hip.normalize(d, _dev(normalizing_mean, dev), _dev(normalizing_range, dev), out=d)

    def save_weights(self, weights_path: str):
        np.save(weights_path, self.cv)

    def save_model(self):
        super().save_model()
        if self.cv is None:
            raise ValueError("No Linear CV weights to save. Please compute the CV before saving the model.")
        f = self.model_output_folder
        self.save_weights(os.path.join(f, "cv_weights.npy"))
        if self.cv_norm_mean is None or self.cv_norm_range is None:
            raise ValueError("CV normalization parameters have not been computed. Cannot save model.")
        np.save(os.path.join(f, "cv_norm_mean.npy"), self.cv_norm_mean)
        np.save(os.path.join(f, "cv_norm_range.npy"), self.cv_norm_range)
        np.save(os.path.join(f, "features_norm_mean.npy"), self.features_norm_mean)
        np.save(os.path.join(f, "features_norm_range.npy"), self.features_norm_range)
        model_path = os.path.join(self.output_path, "model.zip")
        zip_files(model_path, str(f))
        shutil.rmtree(f)
        logger.info(f"Model saved to {model_path}")

    def get_cv_parameters(self):
        return {"cv_name": self.cv_name, "cv_dimension": self.cv_dimension, "features_norm_mode": self.feats_norm_mode,
                "features_norm_mean": self.features_norm_mean, "features_norm_range": self.features_norm_range,
                "cv_stats": self.cv_stats, "weights": self.cv}

    def get_cv_type(self) -> str:
        return "linear"

    def _weights_dev(self, dev):
        return _dev(np.ascontiguousarray(self.cv, dtype=np.float32), dev)

    def project_data(self, data, normalize_data: bool = True) -> torch.Tensor:
        """((x - fmean)/frange) @ W, then (. - cv_mean)/cv_range -- one HBM-bound HIP pass
        (reference :918-972).  Returns a CPU tensor; the caller's data is left untouched."""
        if self.cv is None:
            raise ValueError("CV has not been computed. Cannot project data.")
        if self.cv_norm_mean is None or self.cv_norm_range is None:
            raise ValueError("CV normalization parameters have not been computed. Cannot normalize projected data.")
        dev = _device()
        X = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.asarray(data, dtype=np.float32))
        X = X.to(device=dev, dtype=torch.float32).contiguous()
        kw = {}
        if normalize_data:
            if self.features_norm_mean is None or self.features_norm_range is None:
                raise ValueError("Feature normalization parameters have not been computed. Cannot normalize data.")
            kw = dict(fmean=_dev(self.features_norm_mean, dev), frange=_dev(self.features_norm_range, dev))
        out, _ = hip.project_linear(X, self._weights_dev(dev), cvmean=_dev(self.cv_norm_mean, dev),
                                    cvrange=_dev(self.cv_norm_range, dev), **kw)
        return _to_host(out)

    def normalize_cv(self):
        """min / max of Xn @ W over all training frames -> [-1, 1] (reference :974-991)."""
        if self.training_data is None:
            raise ValueError("Training data not loaded. Cannot compute CV statistics for normalization.")
        _, mm = hip.project_linear(self.training_data, self._weights_dev(self.training_data.device), want_out=False, want_minmax=True)
        mm = reduce_minmax(mm, self.comm).cpu().numpy()
        self.cv_stats = {"min": mm[0], "max": mm[1]}
        self.cv_norm_mean = (mm[1] + mm[0]) / 2
        self.cv_norm_range = (mm[1] - mm[0]) / 2
        self.cv_range = [(-1.0, 1.0)] * self.cv_dimension

    def sensitivity_analysis(self):
        """|W| per feature, ascending, one CSV per CV dimension (reference :993-1047, without
        the plots and the per-atom mapping, which need matplotlib / MDAnalysis)."""
        if self.comm.rank != 0:
            return
        sens = np.abs(self.cv)
        for i in range(sens.shape[1]):
            out = Path(self.sensitivity_output_folder) / f"sensitivity_analysis_{i + 1}"
            out.mkdir(parents=True, exist_ok=True)
            order = np.argsort(sens[:, i])
            pd.DataFrame({"sensitivity": sens[order, i]}, index=list(np.array(self.features_ref_labels)[order])).to_csv(
                out / "sensitivity_analysis.csv")

    # covariance blocks of the (normalised) training matrix, global over ranks
    def _covariances(self, lag: int):
        X = self.training_data
        n_local, F = X.shape
        dev = X.device
        # z = xn - shift with shift ~ column mean of xn keeps the fp32 products well conditioned
        shift = (self.features_stats["mean"].astype(np.float64) - np.asarray(self.features_norm_mean, dtype=np.float64)) / \
            np.asarray(self.features_norm_range, dtype=np.float64)
        # always passed, even when the standardisation already centred the columns (shift ~ 0): with a shift the kernel
        # knows its operand is centred and takes the column sums of z_t from its own fragments (cov.hip), no extra pass
        shift_t = _dev(shift.astype(np.float32), dev)
        Xh, n_pairs_local = append_halo(X, lag, self.comm)
        if n_pairs_local <= 0:
            raise ValueError("not enough frames for the requested lag time")
        raw = self.comm.sum_(hip.lagged_cov_raw(Xh, n_pairs_local, lag, shift_t))
        n_pairs = int(round(self.comm.sum_scalar(n_pairs_local, device=dev)))
        delta, C0, Ct = hip.covariances_from_raw(raw.cpu().numpy(), n_pairs, F)
        return delta, C0, Ct, n_pairs

    def plumed_cv_lines(self) -> str:
        """The linear CV section of the reference's assembler (assembler.py:333-381): one COMBINE per
        normalised feature, one per CV component over them, one per min-max normalised component."""
        text = ""
        feats = list(self.features_ref_labels)
        if self.feats_norm_mode is not None:
            text += "\n# Normalized features\n"
            normalized = []
            for i, feat in enumerate(feats):
                text += plumed_combine(f"feat_{i}", [feat], [1 / self.features_norm_range[i]], [self.features_norm_mean[i]])
                normalized.append(f"feat_{i}")
        else:
            normalized = feats
        text += "\n# Collective variable\n"
        for i in range(self.cv.shape[1]):
            text += plumed_combine(f"{self.cv_name}_{i}", normalized, self.cv[:, i])
        # reference assembler.py:367-370: offset = (min + max) / 2, scale = 2 / (max - min) from cv_stats, in the float32 of
        # the pandas min / max they come from (the '%.17g' text shows every bit); a model loaded from model.zip carries
        # the same two numbers as cv_norm_mean / cv_norm_range
        if self.cv_stats:
            cmin, cmax = np.asarray(self.cv_stats["min"], dtype=np.float32), np.asarray(self.cv_stats["max"], dtype=np.float32)
        else:
            cmin = np.asarray(self.cv_norm_mean, dtype=np.float32) - np.asarray(self.cv_norm_range, dtype=np.float32)
            cmax = np.asarray(self.cv_norm_mean, dtype=np.float32) + np.asarray(self.cv_norm_range, dtype=np.float32)
        offset = (cmin + cmax) / 2
        scale = 2 / (cmax - cmin)
        text += "\n# Normalized Collective variable\n"
        for i in range(self.cv.shape[1]):
            text += plumed_combine(f"norm_{self.cv_name}_{i}", [f"{self.cv_name}_{i}"], [scale[i]], [offset[i]])
        return text

    def plumed_cv_labels(self) -> List[str]:
        return [f"norm_{self.cv_name}_{i}" for i in range(self.cv.shape[1])]


class PCACalculator(LinearCalculator):
    """Principal component analysis (reference :2174-2215): eigenvectors of the covariance of
    the normalised features, covariance accumulated by the FP32-MFMA kernel."""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv_name = "pca"

    def compute_cv(self):
        if self.training_data is None:
            logger.error("No training data available to compute PCA.")
            return
        _, C0, _, _ = self._covariances(lag=0)
        self.cv = linalg.pca_components(C0, self.cv_dimension).astype(np.float32)


class TICACalculator(LinearCalculator):
    """Time-lagged independent component analysis (reference :2217-2267)."""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv_name = "tica"
        self.tica_eigenvalues: Optional[np.ndarray] = None

    def compute_cv(self):
        try:
            _, C0, Ct, _ = self._covariances(lag=int(self.configuration.get("lag_time")))
            evals, evecs = linalg.tica_eigh(C0, Ct, reg=1e-6, n_eig=min(self.cv_dimension, C0.shape[0]))
        except Exception as e:  # the reference logs and skips the CV (:2259-2264)
            logger.error(f"TICA could not be computed. Error message: {e}")
            return
        self.tica_eigenvalues = evals
        self.cv = evecs.astype(np.float32)


class HTICACalculator(LinearCalculator):
    """Hierarchical TICA (reference :2269-2384).  The level-1 covariances are the diagonal blocks
    of the global C0 / Ctau and the level-2 ones are T^T C0 T, T^T Ctau T with
    T = block_diag(level-1 eigenvectors): one data pass in total."""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv_name = "htica"
        self.num_subspaces = self.configuration.get("num_subspaces")
        self.subspaces_dimension = self.configuration.get("subspaces_dimension")

    def compute_cv(self):
        F = self.num_features
        split = F // self.num_subspaces
        if split == 0:
            logger.error(f"Number of subspaces {self.num_subspaces} is larger than number of features {F}. Exiting...")
            return
        try:
            _, C0, Ct, _ = self._covariances(lag=int(self.configuration.get("lag_time")))
            level1 = []
            for a in range(0, F, split):  # torch.split semantics: last chunk may be narrower
                b = min(F, a + split)
                _, ev = linalg.tica_eigh(C0[a:b, a:b], Ct[a:b, a:b], reg=1e-6, n_eig=min(self.subspaces_dimension, b - a))
                level1.append(ev)
            T = linalg.block_diag(level1)
            C0_2 = T.T @ C0 @ T
            Ct_2 = T.T @ Ct @ T
            _, ev2 = linalg.tica_eigh(C0_2, Ct_2, reg=1e-6, n_eig=min(self.cv_dimension, T.shape[1]))
        except Exception as e:
            logger.error(f"TICA could not be computed. Error message: {e}")
            return
        self.cv = (T @ ev2).astype(np.float32)


# ======================================================================================= neural
# torch.optim defaults of the optimisers the HIP engine implements (torch 2.x signatures)
_OPTIMIZERS = {
    "Adam": dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, decoupled_weight_decay=False),
    "AdamW": dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False),
    "SGD": dict(lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False),
    "RMSprop": dict(lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False),
    "Adagrad": dict(lr=1e-2, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10),
    "Adamax": dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
    "NAdam": dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, momentum_decay=4e-3, decoupled_weight_decay=False),
    "RAdam": dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False),
    "Adadelta": dict(lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0),
    "ASGD": dict(lr=1e-2, lambd=1e-4, alpha=0.75, t0=1e6, weight_decay=0.0),
    "Rprop": dict(lr=1e-2, etas=(0.5, 1.2), step_sizes=(1e-6, 50.0)),
}
# torch.optim classes the reference's `getattr(torch.optim, name)` would also accept and the engine does not run:
# LBFGS re-evaluates the loss through a closure several times per step (not a per-element update), SparseAdam rejects
# the dense gradients of these models inside torch itself (the reference's try fails there too)
_OPTIMIZERS_REFUSED = {"LBFGS": "needs a closure that re-evaluates the model several times per step",
                       "SparseAdam": "torch.optim.SparseAdam does not support dense gradients (the reference's try fails as well)",
                       "Adafactor": "not part of the reference's pinned torch 2.1.2 (environment_detailed.yml): its getattr fails there",
                       "Muon": "not part of the reference's pinned torch 2.1.2 (environment_detailed.yml): its getattr fails there"}
_IMPLEMENTATION_SWITCHES = ("foreach", "fused", "capturable", "differentiable")   # no effect on the arithmetic


class _torch_threads:
    """with _torch_threads(k): torch's intra-op thread count capped at k for the block (restored afterwards)."""

    def __init__(self, k: int):
        self.k = k

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.k:
            torch.set_num_threads(self.k)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


class _Batches:
    """The batches of one pass over a DictLoader, without the per-batch objects: kind 'idx' (base = the device index list of
    the pass, batch i = base[i * bs:(i + 1) * bs]) or 'range' (base = first row, batch i = rows base + i * bs ...)."""

    def __init__(self, kind: str, base, n: int, bs: int):
        self.kind, self.base, self.n, self.bs = kind, base, int(n), int(bs)
        self.count = (self.n + self.bs - 1) // self.bs if self.n > 0 else 0

    def __len__(self):
        return self.count

    def size(self, i: int) -> int:
        return min(self.bs, self.n - i * self.bs)

    def full(self) -> int:
        """Number of leading batches of the full size bs."""
        return self.n // self.bs

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.count))]
        if i < 0:
            i += self.count
        if not 0 <= i < self.count:
            raise IndexError(i)
        if self.kind == "idx":
            return ("idx", self.base[i * self.bs:i * self.bs + self.size(i)], self.base)
        return ("range", self.base + i * self.bs, self.size(i))

    def __iter__(self):
        return (self[i] for i in range(self.count))


class _HostLRScheduler:
    """torch.optim.lr_scheduler.<name> (reference :1382-1394, adjusted by :1228-1273) evaluated on the host: the very
    torch class runs over a one-parameter optimiser of the configured type, and after each of its steps the
    learning rate -- and beta1 / momentum, which OneCycleLR and CyclicLR cycle too -- are handed to the HIP engine.
    Stepping follows lightning's lr_scheduler_config: interval 'step' (after every optimiser step) or 'epoch'
    (at the end of every epoch), every `frequency`-th time; ReduceLROnPlateau receives the monitored valid_loss."""

    def __init__(self, name: str, kwargs: Dict, config: Dict, opt_name: str, opt_kwargs: Dict, engine):
        cls = getattr(torch.optim.lr_scheduler, name, None)
        if cls is None:
            logger.error(f"Learning rate scheduler {name} not recognized. Exiting...")
            raise ValueError(f"Learning rate scheduler {name} not recognized.")
        self.opt = getattr(torch.optim, opt_name)([torch.nn.Parameter(torch.zeros(1))], **opt_kwargs)
        self.sched = cls(self.opt, **kwargs)
        self.plateau = isinstance(self.sched, torch.optim.lr_scheduler.ReduceLROnPlateau)
        self.interval = config.get("interval", "epoch")
        self.frequency = max(1, int(config.get("frequency", 1)))
        self.monitor = config.get("monitor", "valid_loss")
        self.engine = engine
        self.count = 0
        self.push()

    def push(self):
        g = self.opt.param_groups[0]
        self.engine.set_lr(float(g["lr"]))
        if "betas" in g:
            self.engine.set_momentum(float(g["betas"][0]))
        elif "momentum" in g:
            self.engine.set_momentum(float(g["momentum"]))

    def lr(self) -> float:
        return float(self.opt.param_groups[0]["lr"])

    def _advance(self, metric=None):
        self.count += 1
        if self.count % self.frequency:
            return
        self.opt.step()   # no gradients: a no-op that keeps torch's call-order check quiet
        if self.plateau:
            if metric is None:   # lightning raises MisconfigurationException here
                raise ValueError(f"ReduceLROnPlateau conditioned on metric {self.monitor} which is not available yet "
                                 "(set check_val_every_n_epoch to 1)")
            self.sched.step(metric)
        else:
            self.sched.step()
        self.push()

    def after_step(self):
        if self.interval == "step":
            self._advance()

    def after_epoch(self, last_valid_loss):
        if self.interval == "epoch":
            self._advance(last_valid_loss)


class NonLinear(CVCalculator):
    """Neural CVs trained by the HIP MLP engine (reference :1049-1921)."""

    model_kind: str = ""

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.training_config: Dict = self.configuration.get("training", {})
        g = self.training_config.get("general", {})
        es = self.training_config.get("early_stopping", {})
        self.optimizer_config: Dict = self.training_config.get("optimizer", {}) or {}
        self.lr_scheduler: Optional[Dict] = self.training_config.get("lr_scheduler", None) or None
        self.lr_scheduler_config: Dict = dict(self.training_config.get("lr_scheduler_config", None) or {})
        self.model_to_save = self.training_config.get("model_to_save", "best")
        self.num_tries: int = g.get("num_tries", 10)
        self.seed: int = g.get("seed", 42)
        self.training_validation_lengths: List = g.get("lengths", [0.8, 0.2])
        self.batch_size: int = g.get("batch_size", 32)
        self.shuffle: bool = g.get("shuffle", True)
        self.random_split: bool = g.get("random_split", True)
        self.max_epochs: int = g.get("max_epochs", 100)
        self.check_val_every_n_epoch: int = g.get("check_val_every_n_epoch", 1)
        self.save_check_every_n_epoch: int = g.get("save_check_every_n_epoch", 5)
        self.early_stop_patience: int = es.get("patience", 20)
        self.early_stop_delta: float = es.get("min_delta", 1e-5)
        self.encoder_config: Dict = copy.deepcopy(self.architecture_config.get("encoder", {}) or {})
        dec = self.architecture_config.get("decoder", {})
        self.decoder_config: Optional[Dict] = copy.deepcopy(dec) if dec is not None else None
        self.encoder_hidden_layers: List[int] = list(self.encoder_config.get("layers", []))
        self.decoder_hidden_layers: List[int] = list((self.decoder_config or self.encoder_config).get("layers", []))
        self.num_training_samples: Optional[int] = None
        self.num_validation_samples: Optional[int] = None
        self.cv_score: Optional[float] = None
        self.tries: int = 0
        self.metrics: Optional[Dict[str, list]] = None
        self.training_metrics_paths: List[str] = []
        self.weights_path: Optional[str] = None
        self.engine: Optional[hip.Mlp] = None
        self.training_normalized: Optional[torch.Tensor] = None    # norm_in applied once (device)
        self.validation_normalized: Optional[torch.Tensor] = None

    # ---- configuration -> layer lists (reference :1155-1219)
    @staticmethod
    def _fit_list(values, n_hidden: int, what: str):
        """One entry per hidden layer.  mlcolvar's FeedForward insists on exactly n entries per option and raises
        otherwise -- which is what the schema's 3-entry defaults do to any `layers` list that is not 3 long
        (the reference's shipped default_config.yml: layers [15, 15], no activation given).  Here a list that is
        too long is cut and one that is too short repeats its last entry, with a warning."""
        values = list(values)
        if len(values) != n_hidden:
            logger.warning(f"'{what}' has {len(values)} entries for {n_hidden} hidden layers; adjusting it.")
            values = (values + [values[-1] if values else None] * n_hidden)[:n_hidden]
        return values

    @classmethod
    def _layer_options(cls, cfg: Dict, n_hidden: int):
        """(activation, dropout, batchnorm) with one entry per Linear: the hidden layers' lists plus the
        last-layer entries, as set_up_encoder_last_layer / set_up_decoder_last_layer append them."""
        act = cls._fit_list(cfg.get("activation", ["leaky_relu"] * n_hidden), n_hidden, "activation")
        drop = cls._fit_list(cfg.get("dropout", [None] * n_hidden), n_hidden, "dropout")
        bn = cls._fit_list(cfg.get("batchnorm", [False] * n_hidden), n_hidden, "batchnorm")
        act.append(cfg.get("last_layer_activation", None))
        drop.append(cfg.get("last_layer_dropout", None))
        bn.append(cfg.get("last_layer_batchnorm", False))
        drop = [float(d) if d else 0.0 for d in drop]
        if any(d < 0.0 or d >= 1.0 for d in drop):
            raise ValueError("dropout probabilities must lie in [0, 1)")
        return act, drop, [bool(b) for b in bn]

    def _optimizer_options(self) -> Tuple[str, Dict]:
        """(torch.optim class name, its keyword arguments with torch's defaults filled in) -- reference :1376-1380."""
        name = self.optimizer_config.get("name", "Adam")
        if name not in _OPTIMIZERS:
            if not hasattr(torch.optim, name):
                raise ValueError(f"Optimizer {name} not recognized.")
            why = _OPTIMIZERS_REFUSED.get(name, "")
            raise NotImplementedError(f"optimizer {name} is not implemented by the HIP engine{': ' + why if why else ''} (have: {sorted(_OPTIMIZERS)})")
        kw = dict(_OPTIMIZERS[name])
        for k, v in (self.optimizer_config.get("kwargs", {}) or {}).items():
            if k in _IMPLEMENTATION_SWITCHES:
                continue
            if k == "maximize":   # every torch.optim class takes it: the update uses the negated gradient
                kw["maximize"] = bool(v)
                continue
            if k not in kw:
                raise TypeError(f"{name}.__init__() got an unexpected keyword argument '{k}'")
            kw[k] = v
        return name, kw

    @staticmethod
    def _engine_optimizer_kwargs(name: str, kw: Dict) -> Dict:
        if name == "Adam" and kw.get("decoupled_weight_decay"):   # newer torch: Adam(decoupled_weight_decay=True) is AdamW's update
            name = "AdamW"
        out = dict(optimizer=name, lr=float(kw["lr"]), weight_decay=float(kw.get("weight_decay", 0.0)))
        if "betas" in kw:
            out["betas"] = (float(kw["betas"][0]), float(kw["betas"][1]))
        for k in ("eps", "momentum", "dampening", "alpha", "lr_decay", "initial_accumulator_value"):
            if k in kw:
                out[k] = float(kw[k])
        for k in ("amsgrad", "nesterov", "centered", "maximize"):
            if k in kw:
                out[k] = bool(kw[k])
        # further constants, in the order of dcv.h's DCV_OPT_* comments
        extra = {"NAdam": [kw.get("momentum_decay", 4e-3), 1.0 if kw.get("decoupled_weight_decay") else 0.0],
                 "RAdam": [0.0, 1.0 if kw.get("decoupled_weight_decay") else 0.0],
                 "Adadelta": [kw.get("rho", 0.9)],
                 "ASGD": [kw.get("lambd", 1e-4), kw.get("alpha", 0.75), kw.get("t0", 1e6)],
                 "Rprop": list(kw.get("etas", (0.5, 1.2))) + list(kw.get("step_sizes", (1e-6, 50.0)))}.get(name)
        if extra is not None:
            out["opt_params"] = [float(v) for v in extra]
        if name == "ASGD":
            out.pop("alpha", None)   # ASGD's alpha is the eta exponent (opt_params), not RMSprop's smoothing constant
        return out

    def _scheduler_options(self, steps_per_epoch: int) -> Optional[Tuple[str, Dict, Dict]]:
        """(name, kwargs, lr_scheduler_config) after adjust_lr_scheduler (reference :1228-1273)."""
        if self.lr_scheduler is None:
            return None
        name = self.lr_scheduler.get("name", "")
        kwargs = dict(self.lr_scheduler.get("kwargs", {}) or {})
        config = dict(self.lr_scheduler_config)
        if name == "OneCycleLR":
            kwargs.setdefault("max_lr", 1e-3)
            kwargs.setdefault("epochs", self.max_epochs)
            kwargs.setdefault("steps_per_epoch", steps_per_epoch)
            config["interval"] = "step"
        elif name == "ReduceLROnPlateau":
            kwargs.setdefault("patience", self.early_stop_patience // 4)
            kwargs.setdefault("cooldown", self.early_stop_patience // 8)
            config["interval"] = "epoch"
        return name, kwargs, config

    # ---- data
    def _norm_tensors(self, dev):
        return _dev(self.features_norm_mean, dev), _dev(self.features_norm_range, dev)

    def set_training_matrix(self, X, feature_names=None, labels=None):
        super().set_training_matrix(X, feature_names, labels)
        dev = self.training_data.device
        # norm_in of the model, applied once: (x - mean)/range is elementwise, so the values are
        # bit-identical to applying it inside the forward pass (reference :1366-1374)
        self.training_normalized = hip.normalize(self.training_data, *self._norm_tensors(dev))
        if self.validation_data is not None:
            self.validation_normalized = hip.normalize(self.validation_data, *self._norm_tensors(dev))

    def set_validation_matrix(self, X):
        super().set_validation_matrix(X)
        if self.features_norm_mean is not None:   # the training statistics normalise the validation frames too
            self.validation_normalized = hip.normalize(self.validation_data, *self._norm_tensors(self.validation_data.device))

    # ---- model description, implemented by the subclasses
    def layer_plan(self) -> Tuple[List[int], List[Optional[str]], List[float], int]:
        """(dims, activation per Linear, dropout per Linear, index of the latent layer)."""
        raise NotImplementedError

    def n_samples_local(self) -> int:
        raise NotImplementedError

    def n_val_samples_local(self) -> int:
        raise NotImplementedError

    def lag(self) -> int:
        return 0

    def check_num_samples(self):
        """reference :1278-1295: counted in FRAMES (len(self.training_data)), also for time-lagged pairs."""
        if self.validation_data is not None:
            self.num_training_samples = self.num_frames_global
            self.num_validation_samples = int(round(self.comm.sum_scalar(self.validation_data.shape[0], device=self.validation_data.device)))
        else:
            self.num_training_samples = int(self.num_frames_global * self.training_validation_lengths[0])
            self.num_validation_samples = self.num_frames_global - self.num_training_samples
        logger.info(f"Number of training samples: {self.num_training_samples}")
        logger.info(f"Number of validation samples: {self.num_validation_samples}")

    def check_batch_size(self) -> int:
        """batch >= number of training samples -> the power of two below it (reference :1297-1309)."""
        if self.batch_size >= self.num_training_samples:
            self.batch_size = closest_power_of_two(self.num_training_samples)
            logger.warning(f"The batch size is larger than the number of samples in the training set. "
                           f"Setting the batch size to the closest power of two: {self.batch_size}")
        return self.batch_size

    @staticmethod
    def _split(n: int, lengths, random_split: bool, generator):
        """DictModule split: fractional lengths = floor + round-robin remainder; random_split uses
        ONE randperm of the given generator, else consecutive blocks."""
        sizes = [int(np.floor(n * f)) for f in lengths]
        for i in range(n - sum(sizes)):
            sizes[i % len(sizes)] += 1
        perm = torch.randperm(n, generator=generator) if random_split else None
        out, off = [], 0
        for s in sizes:
            out.append((perm[off:off + s].clone() if perm is not None else (off, s)))
            off += s
        return out

    def _batches(self, part, batch_size: int, dev):
        """DictLoader: consecutive slices of batch_size, last partial batch kept, fresh randperm per
        epoch when shuffling.  A sequence of ('idx', tensor, whole list) or ('range', row0, count) entries, materialised on
        access (an epoch of 10 000 batches issued by ONE dcv_mlp_train_steps call never builds 10 000 views)."""
        if isinstance(part, tuple):
            off, n = part
            if self.shuffle:
                part = torch.arange(off, off + n)
            else:
                return _Batches("range", off, n, batch_size if batch_size > 0 else n)
        idx = part
        if self.shuffle:
            idx = idx[torch.randperm(len(idx))]   # the reference's permutation (global CPU generator)
        bs = batch_size if batch_size > 0 else len(idx)
        idx_d = idx.to(dev).contiguous()   # one copy per epoch; the batches are consecutive views of it (third field: the whole list)
        return _Batches("idx", idx_d, len(idx), bs)

    @staticmethod
    def _part_len(part) -> int:
        return part[1] if isinstance(part, tuple) else len(part)

    def _init_linears(self, dims):
        """torch.nn.Linear default initialisation in construction order, consuming the global
        RNG exactly as create_model() does."""
        lins = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
        return [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in lins]

    def _step(self, Xn, batch, train: bool, global_batch_of):
        kw = dict(idx=batch[1]) if batch[0] == "idx" else dict(row0=batch[1], batch=batch[2])
        n = int(batch[1].numel()) if batch[0] == "idx" else int(batch[2])
        if not self.comm.active:
            (self.engine.train_step if train else self.engine.eval_step)(Xn, **kw)
            return
        self.engine.data_parallel_step(Xn, self.comm._dist, global_batch_of(n), train=train, group=self.comm.group, **kw)

    def _run_batches(self, Xn, batches, train: bool, global_batch_of, per_step=None):
        """One pass over the loader's batches: a training step (train) or an evaluation step per batch.  On one GPU and with
        nothing to do between steps (`per_step` None: no learning-rate scheduler) the full-sized batches go down in ONE call --
        dcv_mlp_train_steps / dcv_mlp_eval_steps: the epoch loop runs behind the C-ABI, and small networks are evaluated many
        batches per launch -- and the ragged last batch follows; launches, parameters and loss records are those of the
        step-by-step loop, bit for bit."""
        k = 0
        if not self.comm.active and per_step is None and len(batches) > 1:
            k = batches.full()
            if k > 0:
                many = self.engine.train_steps if train else self.engine.eval_steps
                if batches.kind == "idx":
                    many(Xn, batches.bs, k, idx=batches.base)
                else:
                    many(Xn, batches.bs, k, row0=batches.base)
        for b in batches[k:]:
            self._step(Xn, b, train, global_batch_of)
            if per_step is not None:
                per_step()

    def _records_to_metrics(self, rec: np.ndarray, need_eig: bool = True):
        """(weighted mean loss, weighted mean eigenvalues or None, TICA buffers of the last record)."""
        w = rec[:, 1]
        loss = float((rec[:, 0] * w).sum() / w.sum())
        return loss, None, None

    def train(self) -> bool:
        """Multi-try training driver (reference :1456-1553): seed + try, split, model init,
        epochs with validation, early stopping, best / last snapshot; lowest score wins.  As there, a try
        that raises is logged and the next one runs; no valid try -> False (the CV is skipped)."""
        logger.info(f"Training {cv_names_map[self.cv_name]} ...")
        dev = self.training_data.device
        try:
            dims, acts, drops, latent = self.layer_plan()
            opt_name, opt_kw = self._optimizer_options()
        except Exception as e:
            logger.error(f"{cv_names_map[self.cv_name]} cannot be set up: {e}")
            return False
        separate_val = self.validation_data is not None
        if separate_val and self.validation_normalized is None:
            self.validation_normalized = hip.normalize(self.validation_data, *self._norm_tensors(dev))
        n_local = self.n_samples_local()
        n_val_local = self.n_val_samples_local() if separate_val else 0
        if self.comm.active:  # identical batch plans on every rank: use the smallest shard size
            n_local = int(self.comm.min_(torch.tensor([n_local], dtype=torch.int64, device=dev)).item())
            n_val_local = int(self.comm.min_(torch.tensor([n_val_local], dtype=torch.int64, device=dev)).item())
        if n_local < 1 or (separate_val and n_val_local < 1):
            logger.error("Not enough samples to train on.")
            return False
        self.check_num_samples()
        self.check_batch_size()
        world = self.comm.world
        local_bs = max(1, self.batch_size // world) if self.batch_size > 0 else 0
        best = None
        for try_num in range(1, self.num_tries + 1):
            self.tries = try_num
            try:
                # The host side of a try is small torch-CPU work (a randperm and a gather per epoch, the initialisation of a few
                # Linears): with the 128 intra-op threads of a GPU host torch.randperm takes 15 ms for 160 000 indices against
                # 1.3 ms with 8 (its result does not depend on the count: tests/test_host_cpu.py).  Capped once per try --
                # growing the pool back costs ~12 ms, so not per epoch.
                with _torch_threads(8):
                    result = self._train_once(try_num, dims, acts, drops, latent, opt_name, opt_kw, n_local, n_val_local, local_bs, dev)
            except Exception as e:
                logger.error(f"Training try {try_num} failed with an exception: {e}")
                continue
            if result is None:
                continue
            logger.info(f"Try {try_num}/{self.num_tries}: score = {result['score']:.5f}")
            if best is None or result["score"] < best["score"]:
                best = result
                logger.info(f"  -> New best model (try {try_num}).")
        if best is None:
            logger.error(f"{cv_names_map[self.cv_name]} did not produce a valid model after {self.num_tries} tries.")
            return False
        self.cv = best["state"]
        self.cv_score = best["score"]
        self.metrics = best["metrics"]
        self.engine.set_linears(self.cv["linears"], bn=self.cv.get("bn"))
        logger.info(f"Best model score across {self.num_tries} tries: {best['score']:.5f}")
        return True

    def _train_once(self, try_num, dims, acts, drops, latent, opt_name, opt_kw, n_local, n_val_local, local_bs, dev):
        seed = self.seed + try_num
        import random

        random.seed(seed)
        np.random.seed(seed % (2 ** 32))
        gen = torch.manual_seed(seed)           # seed_everything + DictModule(generator=manual_seed(seed))
        separate_val = self.validation_data is not None
        # RNG order (SURVEY.md Appendix A.6 i): create_model() first, the split inside trainer.fit second --
        # unless a scheduler is configured, whose _adjust_lr_scheduler_from_datamodule splits before the model exists
        split_first = self.lr_scheduler is not None and not separate_val
        parts = None
        if split_first:
            parts = self._split(n_local, self.training_validation_lengths, self.random_split, gen)
        linears = self._init_linears(dims)
        if separate_val:       # DictLoader over the whole training set and over the validation set (:1485-1492)
            train_part, val_part = (0, n_local), (0, n_val_local)
        else:
            if parts is None:
                parts = self._split(n_local, self.training_validation_lengths, self.random_split, gen)
            parts = [p if isinstance(p, tuple) or self.shuffle else p.to(dev) for p in parts]  # index plans live on the GPU
            train_part, val_part = parts[0], parts[1]
        Xn_train = self.training_normalized
        Xn_val = self.validation_normalized if separate_val else self.training_normalized
        n_tr, n_va = self._part_len(train_part), self._part_len(val_part)
        bs = local_bs if local_bs > 0 else max(n_tr, n_va, 1)
        if self.engine is not None:
            self.engine.close()
        nmax = max(1, min(bs, max(n_tr, n_va)))
        self.engine = hip.Mlp(self.model_kind, dims, acts, max_batch=nmax, lag=self.lag(), latent_layer=latent,
                              tica_reg=float(self.configuration.get("tica_regularization", 1e-6)), dropout=drops, seed=seed, device=dev,
                              batchnorm=getattr(self, "_bn_plan", None),
                              **self._engine_optimizer_kwargs(opt_name, opt_kw))
        self.engine.set_linears(linears)
        self.engine.set_rank(self.comm.rank)   # data-parallel ranks hold the same seed: independent dropout masks per rank
        if self.comm.world > 1 and any(getattr(self, "_bn_plan", None) or []):
            # each rank would normalise with its local rows and keep its own running statistics: no longer the reference's
            # single-process fit (dcv_mlp_dp_step refuses it too) -- say so before the first step
            raise ValueError("batchnorm / last_layer_batchnorm are not supported in a frame-sharded (multi-GPU) fit: run this CV on one GPU "
                             "or drop the option")
        if self.model_kind == "ae":
            self.engine.set_feature_range(self.features_norm_range)
        self._stats_view = self.engine.stats_view()
        self._grads_view = self.engine.grads_view()
        world = self.comm.world
        gb = lambda n: n * world  # equal shards: every rank runs the same batch sizes
        steps_per_epoch = (n_tr + bs - 1) // bs   # len(train_loader)
        sched = None
        so = self._scheduler_options(steps_per_epoch)
        if so is not None:
            sched = _HostLRScheduler(so[0], so[1], so[2], opt_name, {k: v for k, v in opt_kw.items()}, self.engine)
        metrics: Dict[str, list] = {"train_loss": [], "valid_loss": [], "epoch": []}
        if sched is not None:
            metrics["lr"] = []
        best_score, best_state, last_state = float("inf"), None, None
        es_best, wait = float("inf"), 0
        last_valid = None
        next_tb = None
        for epoch in range(self.max_epochs):
            tb = next_tb if next_tb is not None else self._batches(train_part, bs, dev)
            do_val = (epoch + 1) % self.check_val_every_n_epoch == 0
            vb = self._batches(val_part, bs, dev) if do_val else []
            self.engine.reset_log(len(tb) + len(vb))
            # (a scheduler stepped per EPOCH -- lightning's default interval -- has nothing to do between the steps of an epoch)
            self._run_batches(Xn_train, tb, True, gb, per_step=sched.after_step if sched is not None and sched.interval == "step" else None)
            self._run_batches(Xn_val, vb, False, gb)
            # While the device works through the epoch just enqueued: the NEXT epoch's permutation and index upload (a CPU
            # randperm of 10 M indices is ~0.1 s -- as long as the epoch itself).  The draws keep the reference's order (train
            # loader, validation loader, next train loader ...); if this epoch turns out to be the last one, the generator is put
            # back where the reference's would be.
            rng_before_prefetch = torch.get_rng_state() if self.shuffle else None
            next_tb = self._batches(train_part, bs, dev) if self.shuffle and epoch + 1 < self.max_epochs else None
            rec = self.engine.read_log()   # the only host sync of the epoch
            if not np.all(np.isfinite(rec[:, 0])):
                raise FloatingPointError("non-finite loss (ill-conditioned batch covariance?)")
            train_loss, _, _ = self._records_to_metrics(rec[:len(tb)], need_eig=False)   # (only validation logs eigenvalues)
            stop = False
            if do_val:
                valid_loss, eig, buffers = self._records_to_metrics(rec[len(tb):])
                last_valid = valid_loss
                metrics["train_loss"].append(train_loss)
                metrics["valid_loss"].append(valid_loss)
                metrics["epoch"].append(epoch)
                if sched is not None:
                    metrics["lr"].append(sched.lr())
                if eig is not None:
                    for i, v in enumerate(eig):
                        metrics.setdefault(f"valid_eigval_{i + 1}", []).append(float(v))
                if (epoch + 1) % self.save_check_every_n_epoch == 0:  # ModelCheckpoint(save_top_k=1, save_last=True)
                    state = {"linears": self.engine.get_linears(), "tica": buffers, "dims": dims, "acts": acts, "drops": drops, "latent": latent,
                             "bn": self.engine.get_bn() if any(getattr(self, "_bn_plan", None) or []) else None}
                    last_state = state
                    if valid_loss < best_score:
                        best_score, best_state = valid_loss, state
                if valid_loss < es_best - self.early_stop_delta:  # EarlyStopping(monitor=valid_loss, mode=min)
                    es_best, wait = valid_loss, 0
                else:
                    wait += 1
                    stop = wait >= self.early_stop_patience
            if sched is not None:
                sched.after_epoch(last_valid)
            if stop:
                if next_tb is not None:
                    torch.set_rng_state(rng_before_prefetch)
                break
        if metrics["valid_loss"] and min(metrics["valid_loss"]) > metrics["valid_loss"][0]:
            logger.warning(f"Try {try_num}: validation loss did not decrease during training.")
        # _finalize_training (reference :1555-1642): "best" = the checkpoint with the lowest monitored loss and that
        # loss; "last" = the last checkpoint written, scored with the FINAL validation loss of the run
        if self.model_to_save == "best" and best_state is not None:
            state, score = best_state, best_score
        elif last_state is not None:
            state, score = last_state, metrics["valid_loss"][-1]
        else:
            logger.error("Training finished, but no valid model checkpoint was found.")
            return None
        if self.cv_name == "deep_tica" and score < -float(self.cv_dimension):
            logger.warning(f"Deep TICA validation loss ({score:.5f}) is below the theoretical minimum "
                           f"({-float(self.cv_dimension):.5f}). Try reducing the learning rate or increasing 'tica_regularization'.")
            return None
        return {"state": state, "score": score, "metrics": metrics}

    def compute_cv(self):
        if self.train():
            if self.comm.rank == 0:
                self.plot_training_metrics()

    def plot_training_metrics(self):
        """train_loss / valid_loss / epoch .npy -> training_metrics.zip, model_score.txt,
        eigenvalues.txt (reference :1658-1733, 2592-2627; no figures)."""
        if not self.training_config.get("save_loss", True) or self.metrics is None:
            return
        paths = []
        for key in ("train_loss", "valid_loss", "epoch"):
            p = os.path.join(self.training_output_folder, f"{key}.npy")
            np.save(p, np.array(self.metrics[key]))
            paths.append(p)
        np.savetxt(os.path.join(self.training_output_folder, "model_score.txt"), np.array([self.cv_score]), fmt="%.7g")
        if "valid_eigval_1" in self.metrics and self.cv_score in self.metrics["valid_loss"]:
            bi = self.metrics["valid_loss"].index(self.cv_score)
            ev = [self.metrics[f"valid_eigval_{i + 1}"][bi] for i in range(self.cv_dimension)]
            np.savetxt(os.path.join(self.training_output_folder, "eigenvalues.txt"), np.array(ev), fmt="%.7g")
        zip_files(os.path.join(self.training_output_folder, "training_metrics.zip"), *paths)
        remove_files(*paths)

    # ---- projection
    def _infer(self, Xn: torch.Tensor, with_post: bool, want_out=True, want_minmax=False):
        st = self.cv
        dev = Xn.device
        kw = {}
        if st.get("tica") is not None:
            kw.update(tmean=_dev(st["tica"][0], dev), tevecs=_dev(np.ascontiguousarray(st["tica"][1]), dev))
        if with_post and st.get("post") is not None:
            kw.update(pmean=_dev(st["post"][0], dev), prange=_dev(st["post"][1], dev))
        return self.engine.infer(Xn, want_out=want_out, want_minmax=want_minmax, **kw)

    def normalize_cv(self):
        """min / max of the raw CV over the training samples -> Normalization(mode='min_max') as
        postprocessing (reference :1735-1754)."""
        rows = self.n_samples_local()
        _, mm = self._infer(self.training_normalized[:rows], with_post=False, want_out=False, want_minmax=True)
        mm = reduce_minmax(mm, self.comm).cpu().numpy()
        self.cv["post"] = ((mm[1] + mm[0]) / 2.0, (mm[1] - mm[0]) / 2.0)
        self.cv_range = [(-1.0, 1.0)] * self.cv_dimension

    def project_data(self, data, normalize_data: bool = True) -> torch.Tensor:
        """model(x) for every row (reference :1842-1891).  norm_in is part of the model, so
        `normalize_data` is ignored as in the reference."""
        if self.cv is None:
            raise ValueError("No collective variable model to project data.")
        logger.info(f"Projecting data onto {cv_names_map[self.cv_name]} ...")
        dev = _device()
        if data is self.training_data and self.training_normalized is not None:
            Xn = self.training_normalized
        else:
            X = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.asarray(data, dtype=np.float32))
            X = X.to(device=dev, dtype=torch.float32).contiguous()
            if self.cv.get("norm_in") is not None:
                Xn = hip.normalize(X, _dev(self.cv["norm_in"][0], dev), _dev(self.cv["norm_in"][1], dev))
            else:
                Xn = X
        out, _ = self._infer(Xn, with_post=True)
        return _to_host(out)

    # ---- sensitivity
    def summed_cv_gradient(self) -> np.ndarray:
        """d(sum_j cv_j)/d(network output): the layers behind the network are affine, so this is one
        vector for every frame.  y = (h - tmean) @ evecs (Deep-TICA), cv = (y - pmean) / prange."""
        st = self.cv
        d_lat = st["dims"][st["latent"]]
        inv_range = np.ones(self.cv_dimension if st.get("tica") is not None else d_lat, dtype=np.float64)
        if st.get("post") is not None:
            inv_range = 1.0 / np.asarray(st["post"][1], dtype=np.float64)
        if st.get("tica") is not None:
            return np.asarray(st["tica"][1], dtype=np.float64) @ inv_range
        return inv_range

    def sensitivity_scores(self) -> np.ndarray:
        """mlcolvar.explain.sensitivity_analysis(model, dataset, metric='mean_abs_val') restated
        (reference call site :1903): s_i = mean_r |d(sum_j cv_j)/dx_i(x_r)| * std_i over the samples
        of dataset['data'] (x_t rows for Deep-TICA), normalised to sum to one.  The model takes raw
        features (norm_in is its first layer), so d/dx_i = d/dxn_i / norm_range_i.  One input-gradient
        pass of the HIP engine over the resident normalised matrix."""
        rows = self.n_samples_local()
        dev = self.training_normalized.device
        raw = hip.col_stats_raw(self.training_data[:rows] if self.training_data.is_cuda else self.training_data[:rows].to(dev))
        n_glob = int(self.comm.sum_scalar(float(rows), device=dev))
        std = hip.finalize_stats(reduce_col_stats(raw, self.comm), n_glob)["std"].astype(np.float64)
        scale = std.copy()
        if self.cv.get("norm_in") is not None:
            scale = scale / np.asarray(self.cv["norm_in"][1], dtype=np.float64)
        g = self.summed_cv_gradient()
        acc = self.engine.input_sensitivity(self.training_normalized[:rows], _dev(g.astype(np.float32), dev), _dev(scale.astype(np.float32), dev))
        acc = self.comm.sum_(acc).cpu().numpy() if self.comm.active else acc.cpu().numpy()
        score = acc / float(n_glob)
        return score / score.sum()

    def sensitivity_analysis(self):
        """sensitivity_analysis.csv as NonLinear.sensitivity_analysis writes it (reference :1893-1921):
        features ascending by sensitivity.  Plots and the per-atom mapping need matplotlib / MDAnalysis."""
        if self.cv is None or self.engine is None or self.training_normalized is None:
            return
        score = self.sensitivity_scores()
        if self.comm.rank != 0:
            return
        order = np.argsort(score)
        Path(self.sensitivity_output_folder).mkdir(parents=True, exist_ok=True)
        pd.DataFrame({"sensitivity": score[order]}, index=list(np.array(self.features_ref_labels)[order])).to_csv(
            os.path.join(self.sensitivity_output_folder, "sensitivity_analysis.csv"))

    # ---- persistence
    def _export_dropout(self, cfg: Dict, drops: List[float]) -> List[Optional[float]]:
        """Dropout entry per Linear of the exported module tree: mlcolvar adds a Dropout module wherever the
        configured value is not None -- also for 0 (the bundled models carry Dropout(p=0))."""
        n = len(drops)
        given = self._fit_list((cfg or {}).get("dropout", [None] * (n - 1)), n - 1, "dropout") + [(cfg or {}).get("last_layer_dropout", None)]
        return [None if g is None else float(d) for g, d in zip(given, drops)]

    def to_torch_module(self) -> torch.nn.Module:
        raise NotImplementedError

    def save_weights(self, weights_path: str):
        export.save_torchscript(self.to_torch_module(), self.num_features, weights_path)
        self.weights_path = weights_path

    def save_model(self):
        super().save_model()
        if self.cv is None:
            logger.error("No collective variable model to save.")
            return
        self.save_weights(os.path.join(self.model_output_folder, "cv_weights.pt"))
        model_path = os.path.join(self.output_path, "model.zip")
        zip_files(model_path, str(self.model_output_folder))
        shutil.rmtree(self.model_output_folder)
        logger.info(f"Model saved to {model_path}")

    def _load_from_folder(self, folder_path: str):
        super()._load_from_folder(folder_path)
        weights_path = os.path.join(self.model_output_folder, "cv_weights.pt")
        if not os.path.exists(weights_path):
            raise FileNotFoundError(f"CV model weights not found at {weights_path}")
        parts = export.read_torchscript(weights_path)
        self.weights_path = weights_path
        dims = [parts["linears"][0][0].shape[1]] + [w.shape[0] for w, _ in parts["linears"]]
        self.cv = {"linears": parts["linears"], "acts": parts["acts"], "dims": dims, "tica": parts["tica"],
                   "post": parts["postprocessing"], "norm_in": parts["norm_in"], "latent": len(parts["linears"])}
        # projection only needs the chain up to the CV (the AE file's encoder): a plain Linear chain
        bn = parts.get("bn")
        has_bn = bn is not None and any(b is not None for b in bn)
        self.cv["bn"] = bn if has_bn else None
        self.engine = hip.Mlp("deep_tica", dims, parts["acts"], max_batch=32768, lag=0, device=_device(),
                              batchnorm=[b is not None for b in bn] if has_bn else None)
        self.engine.set_linears(parts["linears"], bn=bn if has_bn else None)

    def get_cv_parameters(self):
        return {"cv_name": self.cv_name, "cv_dimension": self.cv_dimension, "weights_path": self.weights_path}

    def get_cv_type(self) -> str:
        return "non-linear"

    def plumed_cv_lines(self) -> str:
        """The non-linear CV section (assembler.py:408-422): normalisations live inside the TorchScript model."""
        return ("\n# Collective variable\n" + f"{self.cv_name}: PYTORCH_MODEL FILE={os.path.abspath(self.weights_path)} "
                f"ARG={','.join(self.features_ref_labels)}\n")

    def plumed_cv_labels(self) -> List[str]:
        return [f"{self.cv_name}.node-{i}" for i in range(self.cv_dimension)]


class AECalculator(NonLinear):
    """Autoencoder CV (reference :2386-2505): encoder [F]+layers+[d], decoder [d]+layers+[F],
    loss = mean squared reconstruction error in the original feature units."""

    model_kind = "ae"

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv_name = "ae"

    def layer_plan(self):
        enc_act, enc_drop, enc_bn = self._layer_options(self.encoder_config, len(self.encoder_hidden_layers))
        dec_cfg = self.decoder_config if self.decoder_config is not None else self.encoder_config
        dec_act, dec_drop, dec_bn = self._layer_options(dec_cfg, len(self.decoder_hidden_layers))
        self._bn_plan = enc_bn + dec_bn
        # the decoder output must cover the range of the normalised features (reference :1188-1207)
        if self.feats_norm_mode == "min_max_range1" and dec_act[-1] != "custom_sigmoid":
            logger.warning(f"The last layer activation function of the decoder is set to {dec_act[-1]}, but the features are "
                           "normalized using min max with range [0, 1]. Changing the activation function to 'sigmoid'.")
            dec_act[-1] = "custom_sigmoid"
        elif self.feats_norm_mode == "min_max_range2" and dec_act[-1] != "tanh":
            logger.warning(f"The last layer activation function of the decoder is set to {dec_act[-1]}, but the features are "
                           "normalized using min max with range [-1, 1]. Changing the activation function to 'tanh'.")
            dec_act[-1] = "tanh"
        if dec_drop[-1]:
            logger.warning("Dropout in the last layer of the decoder is not recommended.")
        enc = [self.num_features] + self.encoder_hidden_layers + [self.cv_dimension]
        dec = [self.cv_dimension] + self.decoder_hidden_layers + [self.num_features]
        return enc + dec[1:], enc_act + dec_act, enc_drop + dec_drop, len(enc) - 1

    def n_samples_local(self) -> int:
        return self.training_data.shape[0]

    def n_val_samples_local(self) -> int:
        return self.validation_data.shape[0]

    def _records_to_metrics(self, rec, need_eig: bool = True):
        w = rec[:, 1]
        return float((rec[:, 0] * w).sum() / w.sum()), None, None

    def to_torch_module(self):
        st = self.cv
        L = st["latent"]
        ne = len(self.encoder_hidden_layers) + 1
        drops = st.get("drops") or [0.0] * len(st["linears"])
        bn = st.get("bn") or [None] * len(st["linears"])
        enc = export.FeedForward(st["linears"][:L], st["acts"][:L], self._export_dropout(self.encoder_config, drops[:L]), bn[:L])
        dec_cfg = self.decoder_config if self.decoder_config is not None else self.encoder_config
        dec = export.FeedForward(st["linears"][L:], st["acts"][L:], self._export_dropout(dec_cfg, drops[L:]), bn[L:])
        norm = export.Normalization(self.features_norm_mean, self.features_norm_range) if self.feats_norm_mode is not None else None
        post = export.Normalization(*st["post"]) if st.get("post") is not None else None
        return export.AutoEncoderCV(norm, enc, dec, post)

    def train(self) -> bool:
        ok = super().train()
        if ok:
            self.cv["norm_in"] = (self.features_norm_mean, self.features_norm_range)
        return ok


class DeepTICACalculator(NonLinear):
    """Deep-TICA CV (reference :2507-2627): norm_in -> FeedForward -> TICA on time-lagged pairs,
    loss = -sum(eigenvalues^2) of the batch TICA."""

    model_kind = "deep_tica"

    def __init__(self, configuration: Optional[Dict] = None, output_path: Optional[str] = None):
        super().__init__(configuration, output_path)
        self.cv_name = "deep_tica"

    def lag(self) -> int:
        return int(self.configuration.get("lag_time"))

    def layer_plan(self):
        act, drop, bn = self._layer_options(self.encoder_config, len(self.encoder_hidden_layers))
        self._bn_plan = bn
        dims = [self.num_features] + self.encoder_hidden_layers + [self.cv_dimension]
        return dims, act, drop, len(dims) - 1

    def n_samples_local(self) -> int:
        return self.training_data.shape[0] - self.lag()   # pairs (i, i+lag) inside this rank's block

    def n_val_samples_local(self) -> int:
        return self.validation_data.shape[0] - self.lag()  # create_timelagged_dataset(validation_data) (reference :2546-2553)

    def _records_to_metrics(self, rec, need_eig: bool = True):
        d = self.cv_dimension
        w = rec[:, 1]
        loss = float((rec[:, 0] * w).sum() / w.sum())
        if not need_eig:
            return loss, None, None
        reg = float(self.configuration.get("tica_regularization", 1e-6))
        # the d x d TICA of every record in one stacked pass (the same LAPACK routine per record as linalg.tica_eigh)
        ev, evecs = linalg.tica_eigh_stack(rec[:, 2:2 + d * d].reshape(-1, d, d), rec[:, 2 + d * d:2 + 2 * d * d].reshape(-1, d, d), reg=reg)
        buffers = (rec[-1, 2 + 2 * d * d:2 + 2 * d * d + d].astype(np.float32), evecs[-1].astype(np.float32))
        eig = (ev * w[:, None]).sum(axis=0) / w.sum()
        return loss, eig, buffers   # buffers: TICA of the LAST batch (SURVEY.md Appendix A.6 ii)

    def to_torch_module(self):
        st = self.cv
        nl = len(st["linears"])
        nn_ = export.FeedForward(st["linears"], st["acts"], self._export_dropout(self.encoder_config, st.get("drops") or [0.0] * nl), st.get("bn"))
        norm = export.Normalization(self.features_norm_mean, self.features_norm_range) if self.feats_norm_mode is not None else None
        tica = export.TICA(st["tica"][1], st["tica"][0])
        post = export.Normalization(*st["post"]) if st.get("post") is not None else None
        return export.DeepTICA(norm, nn_, tica, post)

    def train(self) -> bool:
        ok = super().train()
        if ok:
            self.cv["norm_in"] = (self.features_norm_mean, self.features_norm_range)
        return ok


cv_calculators_map = {
    "pca": PCACalculator,
    "ae": AECalculator,
    "tica": TICACalculator,
    "htica": HTICACalculator,
    "deep_tica": DeepTICACalculator,
}
cv_names_map = {"pca": "PCA", "ae": "AE", "tica": "TICA", "htica": "HTICA", "deep_tica": "DeepTICA", "vae": "VAE", "umap": "UMAP"}
cv_components_map = {"pca": "PC", "ae": "AE", "tica": "TIC", "htica": "HTIC", "deep_tica": "DeepTIC", "vae": "VAE", "umap": "UMAP"}
