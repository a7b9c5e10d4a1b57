"""Configuration and file helpers with the semantics of the reference's
deep_cartograph/modules/common/common.py (validate_configuration :195-232,
merge_configurations :234-259, zip/unzip :72-155, closest_power_of_two :645-666)."""
from __future__ import annotations

import logging
import os
import sys
import zipfile
from typing import Any, Dict, List, Optional, Type

import yaml

logger = logging.getLogger(__name__)


def files_exist(*paths: str) -> bool:
    return all(os.path.isfile(p) for p in paths)


def read_configuration(configuration_path: str) -> Dict[str, Any]:
    """YAML file -> dict; exits when the file is missing (common.py:170-193)."""
    if not files_exist(configuration_path):
        logger.error(f"Configuration file {configuration_path} not found")
        sys.exit(1)
    with open(configuration_path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def validate_configuration(configuration: Dict[str, Any], schema: Type, output_folder: Optional[str]) -> Dict[str, Any]:
    """schema(**configuration).model_dump(), dumped to <output_folder>/configuration.yml."""
    from pydantic import ValidationError

    try:
        validated = schema(**configuration).model_dump()
    except ValidationError as e:
        logger.error(f"Configuration file is not valid: {e}")
        sys.exit(1)
    if output_folder is not None:
        os.makedirs(output_folder, exist_ok=True)
        with open(os.path.join(output_folder, "configuration.yml"), "w") as f:
            yaml.dump(validated, f)
    return validated


def merge_configurations(common_config: Dict, specific_config: Optional[Dict]) -> Dict:
    """Recursive merge: keys of the CV-specific block override the common block, nested dicts
    are merged key by key, everything else in `common` survives."""
    merged = dict(common_config)
    for key, value in (specific_config or {}).items():
        if isinstance(merged.get(key), dict) and isinstance(value, dict):
            merged[key] = merge_configurations(merged[key], value)
        else:
            merged[key] = value
    return merged


def closest_power_of_two(n: int) -> int:
    """Closest power of two strictly below n: 2**floor(log2 n), halved when n is itself a power of
    two (reference common.py:645-666; clamps the batch size, cv_calculator.py:1306-1307).  The
    reference yields 0.5 for n = 1; 1 is returned here (a batch cannot be smaller)."""
    if n <= 1:
        return 1
    p = 1 << (int(n).bit_length() - 1)
    return p // 2 if p == n else p


def zip_files(output_zip_path: str, *paths: str) -> None:
    """Files go to the archive root; a directory keeps its own name as the top-level folder
    (so `model/` inside model.zip), as the reference's zip_files does."""
    with zipfile.ZipFile(output_zip_path, "w", zipfile.ZIP_DEFLATED) as z:
        for path in paths:
            if os.path.isfile(path):
                z.write(path, arcname=os.path.basename(path))
            elif os.path.isdir(path):
                parent = os.path.dirname(os.path.normpath(path))
                for root, _, files in os.walk(path):
                    for name in sorted(files):
                        full = os.path.join(root, name)
                        z.write(full, arcname=os.path.relpath(full, parent))
            else:
                logger.warning(f"Skipped: path '{path}' does not exist.")


def unzip_files(zip_path: str, output_folder: str) -> None:
    if not os.path.isfile(zip_path):
        logger.error(f"ZIP file '{zip_path}' does not exist.")
        return
    os.makedirs(output_folder, exist_ok=True)
    with zipfile.ZipFile(zip_path, "r") as z:
        z.extractall(output_folder)


def remove_files(*paths: str) -> None:
    for p in paths:
        if os.path.isfile(p):
            os.remove(p)


def read_features_list(features_path: Optional[str]) -> Optional[List[str]]:
    """One feature name per line, or None to use every column."""
    if features_path is None:
        return None
    with open(features_path) as f:
        return [line.strip() for line in f if line.strip()]


def get_unique_path(path: str) -> str:
    """path, or path_1, path_2 ... if it already exists."""
    if not os.path.exists(path):
        return path
    i = 1
    while os.path.exists(f"{path}_{i}"):
        i += 1
    return f"{path}_{i}"
