"""ctypes binding of libdcv.so (include/dcv.h).  Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdcv.so")

DCV_MAX_LAYERS = 16
ACT = {None: 0, "linear": 0, "leaky_relu": 1, "relu": 2, "tanh": 3, "elu": 4, "softplus": 5, "shifted_softplus": 6, "custom_sigmoid": 7}
OPTIMIZER = {"Adam": 0, "AdamW": 1, "SGD": 2, "RMSprop": 3, "Adagrad": 4, "Adamax": 5, "NAdam": 6, "RAdam": 7, "Adadelta": 8, "ASGD": 9, "Rprop": 10}
MODEL_DEEPTICA = 1
MODEL_AE = 2


class MlpDesc(C.Structure):
    _fields_ = [
        ("model", C.c_int32),
        ("n_layers", C.c_int32),
        ("dims", C.c_int32 * (DCV_MAX_LAYERS + 1)),
        ("act", C.c_int32 * DCV_MAX_LAYERS),
        ("latent_layer", C.c_int32),
        ("lag", C.c_int32),
        ("max_batch", C.c_int32),
        ("tica_reg", C.c_double),
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("optimizer", C.c_int32),
        ("amsgrad", C.c_int32),
        ("nesterov", C.c_int32),
        ("centered", C.c_int32),
        ("momentum", C.c_double),
        ("dampening", C.c_double),
        ("alpha", C.c_double),
        ("lr_decay", C.c_double),
        ("initial_accumulator_value", C.c_double),
        ("opt_p", C.c_double * 4),
        ("dropout", C.c_float * DCV_MAX_LAYERS),
        ("seed", C.c_uint64),
        ("batchnorm", C.c_int32 * DCV_MAX_LAYERS),
        ("bn_eps", C.c_double),
        ("bn_momentum", C.c_double),
        ("maximize", C.c_int32),
    ]


class DcvError(RuntimeError):
    pass


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_SZ = C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/dcv.h
SIGNATURES = {
    "dcv_abi_version": (C.c_int, []),
    "dcv_last_error": (C.c_char_p, []),
    "dcv_device_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_I64), C.c_char_p, _SZ]),
    "dcv_col_stats_workspace": (_SZ, [_I64, _I32]),
    "dcv_col_stats": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _SZ, _P]),
    "dcv_normalize": (C.c_int, [_P, _P, _I64, _I32, _I64, _I64, _P, _P, _P]),
    "dcv_lagged_cov_workspace": (_SZ, [_I64, _I32, _I32]),
    "dcv_lagged_cov": (C.c_int, [_P, _I64, _I32, _I64, _I32, _P, _P, _P, _SZ, _P]),
    "dcv_project_linear_workspace": (_SZ, [_I64, _I32, _I32]),
    "dcv_project_linear": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "dcv_mlp_create": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(_P)]),
    "dcv_mlp_destroy": (None, [_P]),
    "dcv_mlp_num_params": (_I64, [_P]),
    "dcv_mlp_param_offset": (_I64, [_P, _I32, _I32]),
    "dcv_mlp_params": (_P, [_P]),
    "dcv_mlp_grads": (_P, [_P]),
    "dcv_mlp_set_params": (C.c_int, [_P, _P, _P]),
    "dcv_mlp_get_params": (C.c_int, [_P, _P, _P]),
    "dcv_mlp_set_lr": (C.c_int, [_P, C.c_double]),
    "dcv_mlp_set_momentum": (C.c_int, [_P, C.c_double]),
    "dcv_mlp_dropout_mask": (C.c_int, [_P, _I32, _I64, _I64, _P, _P]),
    "dcv_mlp_dropout_step": (_I64, [_P]),
    "dcv_mlp_last_path": (_I32, [_P]),
    "dcv_gemm_tn_split": (C.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "dcv_mlp_set_row_sharing": (C.c_int, [_P, _I32]),
    "dcv_mlp_set_feature_range": (C.c_int, [_P, _P, _P]),
    "dcv_mlp_forward": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I32, _P]),
    "dcv_mlp_stats": (_P, [_P]),
    "dcv_mlp_stats_len": (_I32, [_P]),
    "dcv_mlp_backward": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I64, _I32, _P]),
    "dcv_mlp_apply": (C.c_int, [_P, _P]),
    "dcv_mlp_set_upper_grads_callback": (C.c_int, [_P, _P, _P]),
    "dcv_mlp_dp_step": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I64, _I32, _I32, _P, _P, _P]),
    "dcv_comm_unique_id": (C.c_int, [_P]),
    "dcv_comm_create": (C.c_int, [_I32, _I32, _P, C.POINTER(_P)]),
    "dcv_comm_destroy": (None, [_P]),
    "dcv_comm_bind_stream": (C.c_int, [_P, _P]),
    "dcv_comm_allreduce": (C.c_int, [_P, _P, _I64, _I32, _I32, _P]),
    "dcv_comm_dp_allreduce_fn": (_P, []),
    "dcv_comm_world": (_I32, [_P]),
    "dcv_comm_rank": (_I32, [_P]),
    "dcv_mlp_set_rank": (C.c_int, [_P, _I32]),
    "dcv_mlp_bn_state": (C.c_int, [_P, _I32, _P, _P, C.POINTER(_I64), _I32, _P]),
    "dcv_mlp_layer_output": (C.c_int, [_P, _I32, _I64, _P, _P]),
    "dcv_mlp_train_step": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _P]),
    "dcv_mlp_eval_step": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _P]),
    "dcv_mlp_eval_steps": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I32, _P]),
    "dcv_mlp_train_steps": (C.c_int, [_P, _P, _I64, _P, _I64, _I32, _I32, _P]),
    "dcv_mlp_log_width": (_I32, [_P]),
    "dcv_mlp_reset_log": (C.c_int, [_P, _I32, _P]),
    "dcv_mlp_read_log": (C.c_int, [_P, _P, _I32, C.POINTER(_I32), _P]),
    "dcv_mlp_profile_begin": (C.c_int, [_P, _I32, _I32]),
    "dcv_mlp_profile_end": (C.c_int, [_P, _P, _P]),
    "dcv_mlp_profile_pause": (C.c_int, [_P, _I32]),
    "dcv_mlp_graph_launches": (_I64, [_P]),
    "dcv_mlp_set_graph": (C.c_int, [_P, _I32]),
    "dcv_mlp_infer": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "dcv_mlp_input_sensitivity_workspace": (_SZ, [_P, _I64]),
    "dcv_mlp_input_sensitivity": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    "dcv_kmeans_workspace": (_SZ, [_I64, _I32, _I32]),
    "dcv_kmeans_step": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P, _P, _P, _P, _SZ, _P]),
    "dcv_kmeanspp_workspace": (_SZ, [_I64, _I32]),
    "dcv_kmeanspp_potentials": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P, _P, _P, _SZ, _P]),
    "dcv_kmeanspp_update": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P, _P, _P, _SZ, _P]),
    "dcv_set_gemm_mode": (C.c_int, [C.c_int]),
    "dcv_get_gemm_mode": (C.c_int, []),
    "dcv_label_stats_workspace": (_SZ, [_I64, _I32, _I32]),
    "dcv_label_stats": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _P, _P, _SZ, _P]),
    "dcv_cluster_dist_sums": (C.c_int, [_P, _I64, _P, _P, _I32, _I32, _P, _P]),
    "dcv_silhouette_sum_workspace": (_SZ, [_I64]),
    "dcv_silhouette_sum": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _SZ, _P]),
    "dcv_linear_binning_workspace": (_SZ, [_I32, _I32]),
    "dcv_linear_binning": (C.c_int, [_P, _I64, _I64, _I32, _P, _P, _P, _I32, _P, _P, _P, _SZ, _P]),
    "dcv_nearest_rows_workspace": (_SZ, [_I64, _I32, _I32]),
    "dcv_nearest_rows": (C.c_int, [_P, _I64, _I32, _P, _I32, _I64, _P, _P, _P, _SZ, _P]),
    "dcv_nearest_point": (C.c_int, [_P, _I64, _P, _I64, _I32, _P, _P]),
    "dcv_gemm_f32": (C.c_int, [_I32, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P]),
}

_lib = None


def load():
    """Load libdcv.so and bind every entry point.  Raises DcvError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcvError(
            f"libdcv.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C deep_cartograph_amd/csrc` -- there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => header / library mismatch
        fn.restype = res
        fn.argtypes = args
    ver = lib.dcv_abi_version()
    if ver != 5:
        raise DcvError(f"libdcv.so ABI version {ver}, expected 5 (rebuild: make -C deep_cartograph_amd/csrc)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dcv_last_error().decode(errors="replace")
        raise DcvError(f"{what} failed (code {rc}): {msg}")
