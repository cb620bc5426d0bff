"""Seeded synthetic stand-ins for the MipNeRF-360 pretrained scenes.

The real datasets (reference README.md:23,25) are not available offline, so every
benchmark / parity workload is produced by the deterministic generators below.
Outputs use the INRIA ply column names the reference reads
(rasterize.py:98-106,355,358; utils.py:21,27): x y z opacity scale_0..2 rot_0..3
f_dc_0..2 f_rest_0..44, all float32.
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple

import numpy as np

PLY_COLUMNS: List[str] = (
    ["x", "y", "z"]
    + [f"f_dc_{i}" for i in range(3)]
    + [f"f_rest_{i}" for i in range(45)]
    + ["opacity"]
    + [f"scale_{i}" for i in range(3)]
    + [f"rot_{i}" for i in range(4)]
)


class Pose(NamedTuple):
    """COLMAP-convention extrinsics: x_cam = R(qvec) @ x_world + tvec, z forward, y down."""

    image_id: int
    qvec: np.ndarray  # (4,) float64, (w, x, y, z)
    tvec: np.ndarray  # (3,) float64
    name: str


def _columns_from_blocks(xyz, log_scale, rot, opacity_logit, f_dc, f_rest) -> Dict[str, np.ndarray]:
    cols: Dict[str, np.ndarray] = {}
    for i, k in enumerate("xyz"):
        cols[k] = np.ascontiguousarray(xyz[:, i], dtype=np.float32)
    for i in range(3):
        cols[f"f_dc_{i}"] = np.ascontiguousarray(f_dc[:, i], dtype=np.float32)
    for i in range(45):
        cols[f"f_rest_{i}"] = np.ascontiguousarray(f_rest[:, i], dtype=np.float32)
    cols["opacity"] = np.ascontiguousarray(opacity_logit, dtype=np.float32)
    for i in range(3):
        cols[f"scale_{i}"] = np.ascontiguousarray(log_scale[:, i], dtype=np.float32)
    for i in range(4):
        cols[f"rot_{i}"] = np.ascontiguousarray(rot[:, i], dtype=np.float32)
    return cols


def _appearance(rng: np.random.Generator, n: int):
    rot = rng.standard_normal((n, 4), dtype=np.float32)
    pick = rng.random(n) < 0.5
    opacity_logit = np.where(
        pick, rng.normal(-2.0, 1.0, n), rng.normal(3.0, 1.5, n)
    ).astype(np.float32)
    f_dc = rng.normal(0.3, 0.9, (n, 3)).astype(np.float32)
    f_rest = (rng.standard_normal((n, 45), dtype=np.float32) * np.float32(0.12)).astype(np.float32)
    return rot, opacity_logit, f_dc, f_rest


def mip360_like(n: int, seed: int) -> Dict[str, np.ndarray]:
    """Object-centric scene: 60 % dense foreground blob + 40 % far background shell."""
    rng = np.random.default_rng(seed)
    n_fg = int(round(0.6 * n))
    n_bg = n - n_fg
    fg = rng.standard_normal((n_fg, 3)) * np.array([1.2, 1.2, 0.6])
    d = rng.standard_normal((n_bg, 3))
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    radius = np.exp(rng.uniform(math.log(8.0), math.log(60.0), n_bg))
    bg = d * radius[:, None]
    base_fg = rng.normal(-5.6, 0.8, n_fg)
    base_bg = np.log(0.0015 * radius) + rng.normal(0.0, 0.6, n_bg)
    base = np.concatenate([base_fg, base_bg])
    log_scale = base[:, None] + rng.normal(0.0, 0.5, (n, 3))
    xyz = np.concatenate([fg, bg], axis=0)
    rot, opacity_logit, f_dc, f_rest = _appearance(rng, n)
    # interleave fg/bg so that any prefix / strided subsample keeps the 60/40 mix
    perm = rng.permutation(n)
    return _columns_from_blocks(
        xyz[perm], log_scale[perm], rot, opacity_logit, f_dc, f_rest
    )


def uniform_box(n: int, seed: int) -> Dict[str, np.ndarray]:
    """HBM stress scene (BASELINE.json configs[4]): uniform gaussians in [-10,10]^3."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-10.0, 10.0, (n, 3))
    log_scale = rng.normal(-4.0, 0.5, (n, 3))
    rot, opacity_logit, f_dc, f_rest = _appearance(rng, n)
    return _columns_from_blocks(xyz, log_scale, rot, opacity_logit, f_dc, f_rest)


def subsample(cols: Dict[str, np.ndarray], n: int) -> Dict[str, np.ndarray]:
    """First-n prefix (the generators already shuffle)."""
    return {k: np.ascontiguousarray(v[:n]) for k, v in cols.items()}


def _rotmat_to_qvec(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0."""
    K = np.array(
        [
            [R[0, 0] - R[1, 1] - R[2, 2], 0.0, 0.0, 0.0],
            [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], 0.0, 0.0],
            [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], 0.0],
            [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]],
        ]
    ) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def look_at_pose(eye, target, image_id: int, name: str, world_up=(0.0, 0.0, 1.0)) -> Pose:
    """COLMAP pose (z forward, y down, x right) of a camera at `eye` looking at `target`."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    up = np.asarray(world_up, dtype=np.float64)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=0)  # rows = camera axes in world coords
    return Pose(image_id, _rotmat_to_qvec(R), -R @ eye, name)


def ring_cameras(count: int = 25, radius: float = 4.0, height: float = 1.0, first_id: int = 1) -> List[Pose]:
    """`count` poses on a ring around the z axis, looking at the origin."""
    poses = []
    for i in range(count):
        th = 2.0 * math.pi * i / count
        eye = (radius * math.cos(th), radius * math.sin(th), height)
        poses.append(look_at_pose(eye, (0.0, 0.0, 0.0), first_id + i, f"cam_{i:03d}.png"))
    return poses


def box_camera() -> Pose:
    """Camera for `uniform_box`: at (0,0,-14) looking down +z."""
    return look_at_pose((0.0, 0.0, -14.0), (0.0, 0.0, 0.0), 1, "cam_000.png", world_up=(0.0, -1.0, 0.0))


def pinhole_focal(width: int, fov_x_deg: float = 60.0) -> float:
    return width / (2.0 * math.tan(math.radians(fov_x_deg) / 2.0))
