"""Feature-matrix ingestion with the semantics of the reference's
deep_cartograph/modules/plumed/colvars.py:322-473 (``create_dataframe_from_files``):
PLUMED ``#! FIELDS`` text -> float32 columns, ``start/stop/stride`` row selection, the default
column filter (no ``labels|time|bias|walker`` columns), ``features_list`` selection and order,
a ``traj_label`` column per file, concatenation in file order.

Besides the reference's text format a binary fast path is accepted:
``<name>.npy`` holding a 2-D float32 matrix (memory-mapped, so a 10M x 512 matrix costs no
parse), with column names in ``<name>.names.txt`` (one per line) or ``f0..f{F-1}``.
Topology-based feature-name translation (Biopython / MDAnalysis) is out of scope: files are
expected to share feature names, as every BASELINE configuration does."""
from __future__ import annotations

import logging
import os
import re
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

logger = logging.getLogger(__name__)

_DEFAULT_FILTER = re.compile(r"^(?!.*labels)^(?!.*time)^(?!.*bias)^(?!.*walker)")


def is_binary_matrix(path: str) -> bool:
    return path.endswith(".npy")


def read_column_names(colvars_path: str, features_only: bool = False) -> List[str]:
    """Column names of a PLUMED COLVAR file (``#! FIELDS a b c``) or of a binary matrix."""
    if is_binary_matrix(colvars_path):
        names_path = colvars_path[:-4] + ".names.txt"
        if os.path.exists(names_path):
            with open(names_path) as f:
                names = [line.strip() for line in f if line.strip()]
        else:
            shape = np.load(colvars_path, mmap_mode="r").shape
            names = [f"f{i}" for i in range(shape[1])]
    else:
        with open(colvars_path) as f:
            names = f.readline().split()[2:]
    if features_only:
        names = [n for n in names if _DEFAULT_FILTER.search(n)]
    return names


def _read_matrix(path: str, start: int, stop: Optional[int], stride: int) -> Tuple[np.ndarray, List[str]]:
    """(rows x all columns float32, names) of one file after the row selection."""
    names = read_column_names(path)
    if is_binary_matrix(path):
        arr = np.load(path, mmap_mode="r")
        if arr.ndim != 2:
            raise ValueError(f"{path}: expected a 2-D matrix, got shape {arr.shape}")
        if arr.shape[1] != len(names):
            raise ValueError(f"{path}: {arr.shape[1]} columns but {len(names)} names")
        return arr[start:stop:stride], names
    return _read_text(path, names)[start:stop:stride], names


PARALLEL_PARSE_MIN_BYTES = 64 << 20   # text files above this size are parsed by a pool of processes


def _parse_text_range(args) -> np.ndarray:
    """Worker: rows of the byte range [begin, end) of a COLVAR text file (the range starts at a line start)."""
    import io

    import pandas as pd

    path, begin, end, ncols = args
    with open(path, "rb") as f:
        f.seek(begin)
        blob = f.read(end - begin)
    if not blob.strip():
        return np.empty((0, ncols), dtype=np.float32)
    df = pd.read_csv(io.BytesIO(blob), sep=r"\s+", dtype=np.float32, comment="#", header=None, names=list(range(ncols)))
    return df.to_numpy(dtype=np.float32)


def _line_aligned_ranges(path: str, parts: int):
    """Cut the file into `parts` byte ranges that begin at line starts (rows keep their order)."""
    size = os.path.getsize(path)
    cuts = [0]
    with open(path, "rb") as f:
        for i in range(1, parts):
            f.seek(max(cuts[-1], i * size // parts))
            f.readline()               # finish the line the cut fell into
            pos = f.tell()
            if pos >= size:
                break
            if pos > cuts[-1]:
                cuts.append(pos)
    cuts.append(size)
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]


def _read_text(path: str, names: Sequence[str], workers: Optional[int] = None, min_bytes: Optional[int] = None) -> np.ndarray:
    """All rows x all columns (float32) of a PLUMED COLVAR text file.  The reference parses with one
    `pd.read_csv(sep='\\s+')` (colvars.py:383); at 10M x 512 that is ~50 GB of text, so large files are cut into
    line-aligned byte ranges parsed by a process pool (same parser, same values, rows in file order)."""
    import pandas as pd

    min_bytes = PARALLEL_PARSE_MIN_BYTES if min_bytes is None else min_bytes
    workers = workers or min(16, os.cpu_count() or 1)
    if workers <= 1 or os.path.getsize(path) < min_bytes:
        return pd.read_csv(path, sep=r"\s+", dtype=np.float32, comment="#", header=None, names=list(names)).to_numpy(dtype=np.float32)
    from concurrent.futures import ProcessPoolExecutor

    ranges = _line_aligned_ranges(path, workers * 4)
    with ProcessPoolExecutor(max_workers=workers) as pool:
        blocks = list(pool.map(_parse_text_range, [(path, b, e, len(names)) for b, e in ranges]))
    return np.concatenate(blocks, axis=0)


def load_feature_matrix(colvars_paths: Union[str, Sequence[str]], features_list: Optional[Sequence[str]] = None,
                        start: int = 0, stop: Optional[int] = None, stride: int = 1,
                        shard: Optional[Tuple[int, int]] = None) -> Tuple[np.ndarray, List[str], np.ndarray]:
    """(X float32 C-contiguous [frames, features], feature names, traj_label per frame).

    Same selection rules as ``create_dataframe_from_files``; raises on NaNs in a file, on a
    missing requested feature and on files whose columns disagree when no list is given.

    ``shard = (world, rank)``: X holds only this rank's contiguous block of the concatenated frames
    (parallel.shard_bounds; rows outside it are never materialised -- a memory-mapped .npy is sliced, not
    read); the labels returned are still those of ALL frames, so that every rank can split a gathered
    projection by trajectory."""
    if isinstance(colvars_paths, str):
        colvars_paths = [colvars_paths]
    blocks, labels = [], []
    ref_names: Optional[List[str]] = None
    mats = []
    for path in colvars_paths:
        if not os.path.exists(path):
            raise FileNotFoundError(f"Colvars file not found: {path}")
        mats.append(_read_matrix(path, start, stop, stride))
    lo, hi = 0, sum(m.shape[0] for m, _ in mats)
    if shard is not None:
        from .parallel import shard_bounds

        lo, hi = shard_bounds(hi, int(shard[0]), int(shard[1]))
    offset = 0
    for file_index, path in enumerate(colvars_paths):
        mat, names = mats[file_index]
        labels.append(np.full(mat.shape[0], file_index, dtype=np.int64))
        a, b = min(max(lo - offset, 0), mat.shape[0]), min(max(hi - offset, 0), mat.shape[0])
        offset += mat.shape[0]
        mat = mat[a:b]
        keep = [i for i, n in enumerate(names) if _DEFAULT_FILTER.search(n)]
        kept_names = [names[i] for i in keep]
        if features_list:
            missing = set(features_list) - set(kept_names)
            if missing:
                raise ValueError(f"Features {missing} not found in {path}.")
            pos = {n: i for i, n in zip(keep, kept_names)}
            cols = [pos[n] for n in features_list]
            out_names = list(features_list)
        else:
            cols = keep
            out_names = kept_names
        if ref_names is None:
            ref_names = out_names
        elif out_names != ref_names:
            raise ValueError(f"Column names in {path} do not match those in {colvars_paths[0]}. "
                             "Please provide a features_list to filter and reorder the columns.")
        block = np.ascontiguousarray(mat[:, cols], dtype=np.float32)
        if np.isnan(block).any():
            raise ValueError(f"Clean your data! NaNs found in {path}")
        blocks.append(block)
    if not blocks:
        raise ValueError("No colvars files given.")
    X = blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=0)
    if X.shape[0] == 0:
        raise ValueError("The resulting feature matrix is empty.")
    return X, list(ref_names), np.concatenate(labels)


def write_colvars(path: str, X: np.ndarray, names: Sequence[str], time: Optional[np.ndarray] = None) -> None:
    """Write a PLUMED-style COLVAR text file (used by tests and examples)."""
    with open(path, "w") as f:
        f.write("#! FIELDS time " + " ".join(names) + "\n")
        t = np.arange(X.shape[0], dtype=np.float64) if time is None else time
        for i in range(X.shape[0]):
            f.write(" %.6f " % t[i] + " ".join("%.8f" % v for v in X[i]) + "\n")


def write_binary_matrix(path: str, X: np.ndarray, names: Optional[Sequence[str]] = None) -> None:
    """The binary fast-path format: <path>.npy (+ <path stem>.names.txt)."""
    assert path.endswith(".npy")
    np.save(path, np.ascontiguousarray(X, dtype=np.float32))
    if names is not None:
        with open(path[:-4] + ".names.txt", "w") as f:
            f.write("\n".join(names) + "\n")
