"""Configuration contract of the CV-fit path: pydantic models with the field names, defaults
and `extra` policy of the reference's deep_cartograph/yaml_schemas/{train_colvars,traj_cluster,
traj_projection}.py.  tests/test_host_cpu.py checks model_dump() of the defaults against the
reference's own dump (tests/golden/schema_defaults.json)."""
from __future__ import annotations

from typing import List, Literal, Optional, Union

from pydantic import BaseModel, ConfigDict

Activation = Literal["relu", "elu", "tanh", "softplus", "shifted_softplus", "custom_sigmoid", "leaky_relu", "linear"]


class Optimizer(BaseModel):
    name: str = "Adam"
    kwargs: dict = {"lr": 1.0e-04, "weight_decay": 0.0}


class RLScheduler(BaseModel):
    name: str = "OneCycleLR"
    kwargs: dict = {}


class NeuralNetwork(BaseModel):
    layers: List[int] = [64, 32, 16]
    activation: List[Optional[Activation]] = ["leaky_relu", "leaky_relu", "leaky_relu"]
    batchnorm: List[bool] = [False, False, False]
    dropout: List[Optional[float]] = [None, None, None]
    last_layer_activation: Optional[Activation] = None
    last_layer_batchnorm: bool = False
    last_layer_dropout: Optional[float] = None


class Architecture(BaseModel):
    encoder: NeuralNetwork = NeuralNetwork()
    decoder: NeuralNetwork = NeuralNetwork()


class GeneralSettings(BaseModel):
    num_tries: int = 10
    seed: int = 42
    lengths: List[float] = [0.8, 0.2]
    batch_size: int = 32
    max_epochs: int = 1000
    shuffle: bool = False
    random_split: bool = True
    check_val_every_n_epoch: int = 10
    save_check_every_n_epoch: int = 10


class InputColvars(BaseModel):
    start: int = 0
    stop: Union[int, None] = None
    stride: int = 1


class EarlyStopping(BaseModel):
    patience: int = 20
    min_delta: float = 1.0e-05


class KLAnnealing(BaseModel):
    type: Literal["linear", "sigmoid", "cyclical"] = "linear"
    start_beta: float = 1e-06
    max_beta: float = 0.01
    start_epoch: int = 1000
    n_cycles: int = 4
    n_epochs_anneal: int = 5000


class Trainings(BaseModel):
    general: GeneralSettings = GeneralSettings()
    early_stopping: EarlyStopping = EarlyStopping()
    optimizer: Optimizer = Optimizer()
    lr_scheduler: Optional[RLScheduler] = None
    lr_scheduler_config: Optional[dict] = {"interval": "epoch", "monitor": "valid_loss", "frequency": 1}
    kl_annealing: Optional[KLAnnealing] = None
    save_loss: bool = True
    plot_loss: bool = True
    model_to_save: Literal["best", "last"] = "best"


class BiasArgs(BaseModel):
    temperature: float = 300.0
    sigma: float = 0.05
    pace: int = 500
    grid_min: float = -1.0
    grid_max: float = 1.0
    grid_bin: int = 300
    height: float = 1.0
    bias_factor: float = 10.0
    barrier: float = 50.0
    observation_steps: int = 100
    compression_threshold: float = 0.1


class Bias(BaseModel):
    method: Literal["wt_metadynamics", "opes_metad", "opes_metad_explore", "opes_expanded"] = "opes_metad"
    args: BiasArgs = BiasArgs()
    add_rmsd_restraint: bool = False
    align_waypoint_structures: bool = True
    rmsd_restraint_k: float = 5000.0
    rmsd_restraint_eq: float = 0.4


class CommonCollectiveVariable(BaseModel):
    dimension: int = 2
    lag_time: int = 1
    tica_regularization: float = 1.0e-06
    features_normalization: Optional[Literal["mean_std", "min_max_range1", "min_max_range2"]] = None
    input_colvars: InputColvars = InputColvars()
    architecture: Architecture = Architecture()
    training: Trainings = Trainings()
    num_subspaces: int = 10
    subspaces_dimension: int = 5
    n_neighbors: int = 15
    min_dist: float = 0.1
    metric: str = "euclidean"
    bias: Bias = Bias()


class FesFigure(BaseModel):
    compute: bool = True
    save: bool = True
    temperature: int = 300
    bandwidth: float = 0.05
    num_fes_levels: int = 10
    num_bins: int = 150
    max_fes: float = 30


class TrajProjection(BaseModel):
    plot: bool = True
    num_bins: int = 100
    bandwidth: float = 0.25
    alpha: float = 0.8
    cmap: str = "turbo"
    marker_size: int = 5


class Figures(BaseModel):
    fes: FesFigure = FesFigure()
    traj_projection: TrajProjection = TrajProjection()


class TrainColvarsSchema(BaseModel):
    # per-CV override sections (e.g. `deep_tica: {...}`) are extra fields, as in the reference
    model_config = ConfigDict(extra="allow")
    cvs: List[Literal["pca", "ae", "tica", "htica", "deep_tica", "vae", "umap"]] = ["pca", "ae", "tica", "htica", "deep_tica", "vae", "umap"]
    common: CommonCollectiveVariable = CommonCollectiveVariable()
    figures: Figures = Figures()


class ClusterFigures(BaseModel):
    plot: bool = True
    num_bins: int = 100
    bandwidth: float = 0.25
    alpha: float = 0.8
    cmap: str = "turbo"
    marker_size: int = 5


class TrajClusterSchema(BaseModel):
    run: bool = True
    output_structures: Optional[Literal["centroids", "all"]] = "centroids"
    algorithm: Literal["kmeans", "hdbscan", "hierarchical"] = "hierarchical"
    opt_num_clusters: bool = True
    search_interval: List[int] = [3, 10]
    num_clusters: int = 10
    linkage: str = "complete"
    n_init: int = 20
    min_cluster_size: int = 5
    max_cluster_size: Union[int, None] = None
    min_samples: int = 3
    cluster_selection_epsilon: float = 0
    cluster_selection_method: Literal["eom", "leaf"] = "eom"
    figures: ClusterFigures = ClusterFigures()


class TrajProjectionSchema(BaseModel):
    figures: Figures = Figures()
