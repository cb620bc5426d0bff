"""Scene/model adapters: the reference's `utils.py` surface plus the packed layout the HIP library eats.

`read_color_components`  <- reference utils.py:10-31
`read_scene`             <- reference utils.py:34-58
`pack_gaussians`         : the five dense arrays of include/gsr.h (GsrScene) from ply columns
"""
from __future__ import annotations

import os
from typing import Dict, Mapping, Tuple

import numpy as np
import torch

from .data_reader import BaseImage, Camera, read_extrinsics_binary, read_intrinsics_binary

SH_COEFFS = 16  # degree-3 real SH


def _column(src, name: str) -> np.ndarray:
    """ply element / dict / PlyData -> float32 column."""
    if hasattr(src, "elements"):
        src = src.elements[0]
    return np.asarray(src[name], dtype=np.float32)


def sh_columns_to_array(src) -> np.ndarray:
    """[N,16,3] float32 with sh[n,0,c] = f_dc_c and sh[n,k,c] = f_rest_{15c+k-1}, k = 1..15
    (the channel-major f_rest order of INRIA ply files; reference utils.py:21-31)."""
    n = _column(src, "f_dc_0").shape[0]
    sh = np.empty((n, SH_COEFFS, 3), dtype=np.float32)
    for c in range(3):
        sh[:, 0, c] = _column(src, f"f_dc_{c}")
        for k in range(1, SH_COEFFS):
            sh[:, k, c] = _column(src, f"f_rest_{15 * c + k - 1}")
    return sh


def read_color_components(plydata) -> torch.Tensor:
    """Spherical-harmonics coefficients of every gaussian as a [N,16,3] float32 tensor."""
    return torch.from_numpy(sh_columns_to_array(plydata))


def pack_gaussians(src) -> Dict[str, np.ndarray]:
    """ply columns -> the dense, camera-independent arrays the C ABI takes (GsrScene in include/gsr.h).

    means[N,3] (rasterize.py:354-356), log_scales[N,3] and quats[N,4] exactly as stored, i.e. before the
    exp / normalisation of rasterize.py:97-112, opacity_logit[N] before the sigmoid of :358, sh[N,16,3].
    """
    return {
        "means": np.ascontiguousarray(np.stack([_column(src, k) for k in "xyz"], axis=1)),
        "log_scales": np.ascontiguousarray(np.stack([_column(src, f"scale_{i}") for i in range(3)], axis=1)),
        "quats": np.ascontiguousarray(np.stack([_column(src, f"rot_{i}") for i in range(4)], axis=1)),
        "opacity_logit": np.ascontiguousarray(_column(src, "opacity")),
        "sh": sh_columns_to_array(src),
    }


def read_scene(path_to_scene: str) -> Tuple[Dict[int, BaseImage], Dict[int, Camera]]:
    """COLMAP sparse model of a scene directory -> ({image_id: Image}, {camera_id: Camera})."""
    sparse = os.path.join(path_to_scene, "sparse/0")
    cam_extrinsics = read_extrinsics_binary(os.path.join(sparse, "images.bin"))
    cam_intrinsics = read_intrinsics_binary(os.path.join(sparse, "cameras.bin"))
    return cam_extrinsics, cam_intrinsics
