#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); the fixtures it writes
are data (inputs + expected outputs) and are committed, the reference is not.

Method: the reference's `rasterize.py` is imported unmodified
with this repo's PLY reader registered as the missing third-party `plyfile`
module, `run_rasterization.callback(...)` is run end-to-end on synthetic scenes
written to disk (sparse/0/{cameras,images}.bin, images_K/<name>.png,
point_cloud/iteration_30000/point_cloud.ply), the final frame is captured from
the `plt.imshow` call at rasterize.py:471, and every helper on the hot path is
wrapped by a recorder so its inputs/outputs are captured while the reference's
own glue (rasterize.py:347-446) drives it.

Usage:  python tools/make_golden.py [--only f1,f2,f3,f4,f5] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import tempfile
import time
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)
pkg = importlib.import_module("torch-gaussian-splatting-rasterizer_amd")
syn = importlib.import_module("torch-gaussian-splatting-rasterizer_amd.synthetic")
ply = importlib.import_module("torch-gaussian-splatting-rasterizer_amd.ply")
colmap = importlib.import_module("torch-gaussian-splatting-rasterizer_amd.data_reader")


# --------------------------------------------------------------------------- reference import
def import_reference():
    stub = types.ModuleType("plyfile")
    stub.PlyData, stub.PlyElement = ply.PlyData, ply.PlyElement
    sys.modules["plyfile"] = stub
    sys.path.insert(0, REFERENCE)
    import rasterize as ref  # noqa: the reference module, unmodified

    import logging

    logging.getLogger().setLevel(logging.WARNING)  # the reference sets the root logger to NOTSET (rasterize.py:19-23)
    ref.tqdm.tqdm = lambda it, **k: it
    ref.mpimg.imread = lambda p: np.zeros((2, 2, 3))
    return ref


class Recorder:
    """Wraps the reference's helpers; keeps clones of what goes in and out."""

    HELPERS = [
        "get_covariance_matrix_from_mesh",
        "get_world_to_camera_matrix",
        "get_projection_matrix",
        "sh_to_rgb",
        "project_to_camera_space",
        "compute_2d_covariance",
        "compute_covering_bbox",
    ]

    def __init__(self, ref):
        self.ref = ref
        self.rec = {}
        self.draw_order = []
        self.frames = []
        self._orig = {}

    def __enter__(self):
        import torch

        ref = self.ref

        def clone(x):
            if isinstance(x, torch.Tensor):
                return x.detach().clone().cpu().numpy()
            if isinstance(x, np.ndarray):
                return x.copy()
            return x

        def wrap(name):
            fn = getattr(ref, name)
            self._orig[name] = fn

            def inner(*a, **k):
                out = fn(*a, **k)
                self.rec[name] = {"args": [clone(v) for v in a], "out": clone(out)}
                return out

            setattr(ref, name, inner)

        for h in self.HELPERS:
            wrap(h)

        raster = ref.rasterize_gaussian
        self._orig["rasterize_gaussian"] = raster

        def raster_rec(gaussian_index, bboxes, screen, screen_means, sigmas, rgb, opacity_buffer, opacity):
            if not self.draw_order:
                self.rec["blend_inputs"] = {
                    "bboxes": clone(bboxes),
                    "screen_means": clone(screen_means),
                    "sigmas": clone(sigmas),
                    "rgb": clone(rgb),
                    "opacity": clone(opacity),
                }
            self.draw_order.append(int(gaussian_index))
            return raster(gaussian_index, bboxes, screen, screen_means, sigmas, rgb, opacity_buffer, opacity)

        ref.rasterize_gaussian = raster_rec
        self._orig["imshow"] = ref.plt.imshow
        ref.plt.imshow = lambda x, *a, **k: self.frames.append(np.asarray(x).copy())
        return self

    def __exit__(self, *exc):
        for k, v in self._orig.items():
            if k == "imshow":
                self.ref.plt.imshow = v
            else:
                setattr(self.ref, k, v)
        self.ref.plt.close("all")


# --------------------------------------------------------------------------- scenes on disk
def write_scene(root, cols, poses, cam_w, cam_h, fx, fy, img_w, img_h, scale_factor, camera_id=1):
    from PIL import Image

    sparse = os.path.join(root, "scene", "sparse", "0")
    os.makedirs(sparse, exist_ok=True)
    colmap.write_intrinsics_binary(
        os.path.join(sparse, "cameras.bin"),
        [colmap.Camera(id=camera_id, model="PINHOLE", width=cam_w, height=cam_h,
                       params=np.array([fx, fy, cam_w / 2.0, cam_h / 2.0]))],
    )
    colmap.write_extrinsics_binary(os.path.join(sparse, "images.bin"), poses)
    img_dir = os.path.join(root, "scene", f"images_{scale_factor}")
    os.makedirs(img_dir, exist_ok=True)
    for p in poses:
        Image.new("RGB", (img_w, img_h)).save(os.path.join(img_dir, p.name))
    model = os.path.join(root, "model", "point_cloud", "iteration_30000")
    os.makedirs(model, exist_ok=True)
    ply.write_gaussians_ply(os.path.join(model, "point_cloud.ply"), cols)
    return os.path.join(root, "scene"), os.path.join(root, "model")


def run_reference(ref, cols, pose_list, image_id, W, H, fx_full, fy_full, cam_w, cam_h, scale_factor=2):
    with tempfile.TemporaryDirectory() as tmp:
        scene_dir, model_dir = write_scene(tmp, cols, pose_list, cam_w, cam_h, fx_full, fy_full, W, H, scale_factor)
        with Recorder(ref) as r:
            t0 = time.time()
            ref.run_rasterization.callback(
                input_dir=scene_dir, trained_model_path=model_dir, output_path=os.path.join(tmp, "out"),
                scene_index=image_id, scale_factor=scale_factor, generate_video=False,
            )
            dt = time.time() - t0
    return r, dt


def pack_inputs(cols, pose, W, H, fx_full, fy_full, cam_w, cam_h, scale_factor):
    d = {f"ply_{k}": v for k, v in cols.items()}
    d.update(
        qvec=np.asarray(pose.qvec, np.float64), tvec=np.asarray(pose.tvec, np.float64),
        image_id=np.int64(pose.image_id), width=np.int64(W), height=np.int64(H),
        fx_full=np.float64(fx_full), fy_full=np.float64(fy_full),
        cam_width=np.int64(cam_w), cam_height=np.int64(cam_h), scale_factor=np.int64(scale_factor),
    )
    return d


def pack_intermediates(r):
    rec = r.rec
    b = rec["blend_inputs"] if "blend_inputs" in rec else None
    out = {
        "cov3d": rec["get_covariance_matrix_from_mesh"]["out"],
        "w2c_M": rec["get_world_to_camera_matrix"]["out"],          # untransposed M (rasterize.py:59-77)
        "proj_P": rec["get_projection_matrix"]["out"],              # untransposed P (rasterize.py:123-151)
        "sh": rec["sh_to_rgb"]["args"][1],                          # [N,16,3] (utils.py:10-31)
        "w2c_T": rec["sh_to_rgb"]["args"][2],                       # transposed, as passed downstream
        "rgb": rec["sh_to_rgb"]["out"],
        "cam_means": rec["project_to_camera_space"]["out"],
        "cov2d": rec["compute_2d_covariance"]["out"],               # before the cull zeroing at :388
        "cov2d_after_cull": rec["compute_covering_bbox"]["args"][1],
        "screen_means": rec["compute_covering_bbox"]["args"][0],
        "tile_bboxes": rec["compute_covering_bbox"]["out"],
        "tan_fov_x": np.float64(rec["compute_2d_covariance"]["args"][2]),
        "tan_fov_y": np.float64(rec["compute_2d_covariance"]["args"][3]),
        "focals": np.asarray(rec["compute_2d_covariance"]["args"][4], np.float64),
        "draw_order": np.asarray(r.draw_order, np.int64),
    }
    if b is not None:
        out.update(pixel_bboxes=b["bboxes"], sigmas=b["sigmas"], opacity=b["opacity"])
    return out


def fixture_scene(n, seed, scale_shift):
    cols = syn.mip360_like(n, seed)
    for i in range(3):
        cols[f"scale_{i}"] = (cols[f"scale_{i}"] + np.float32(scale_shift)).astype(np.float32)
    return cols


# --------------------------------------------------------------------------- fixtures
def make_f1(ref, out_dir):
    """unit: N=64, 64x48, every intermediate."""
    W, H = 64, 48
    cols = fixture_scene(64, 11, 3.2)
    poses = syn.ring_cameras(5, first_id=3)
    fx = syn.pinhole_focal(W)
    r, dt = run_reference(ref, cols, poses, poses[1].image_id, W, H, 2 * fx, 2 * fx, 2 * W, 2 * H)
    d = pack_inputs(cols, poses[1], W, H, 2 * fx, 2 * fx, 2 * W, 2 * H, 2)
    d.update(pack_intermediates(r))
    d["image"] = r.frames[0].astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "f1_unit.npz"), **d)
    print(f"f1: drawn {len(r.draw_order)}/64 in {dt:.1f}s, image max {d['image'].max():.3f}")


def make_f2(ref, out_dir):
    """small: N=3000, 160x96 — inputs + final image + light intermediates."""
    W, H = 160, 96
    cols = fixture_scene(3000, 4, 2.0)
    poses = syn.ring_cameras(25)
    fx = syn.pinhole_focal(W)
    pose = poses[7]
    r, dt = run_reference(ref, cols, poses, pose.image_id, W, H, 2 * fx, 2 * fx, 2 * W, 2 * H)
    d = pack_inputs(cols, pose, W, H, 2 * fx, 2 * fx, 2 * W, 2 * H, 2)
    mid = pack_intermediates(r)
    for k in ("rgb", "screen_means", "tile_bboxes", "pixel_bboxes", "sigmas", "opacity", "draw_order", "cam_means"):
        d[k] = mid[k]
    d["image"] = r.frames[0].astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "f2_small.npz"), **d)
    print(f"f2: drawn {len(r.draw_order)}/3000 in {dt:.1f}s, image mean {d['image'].mean():.4f}")


def edge_columns():
    """Hand-built gaussians for the edge cases of SURVEY.md §8(c) F3 (camera at origin looking +z)."""
    rows = []

    def add(xyz, log_scale, rot=(1, 0, 0, 0), opacity=2.0, dc=(0.5, 0.1, -0.3)):
        rows.append((xyz, log_scale, rot, opacity, dc))

    add((0.0, 0.0, 3.0), (-2.0, -2.0, -2.0))                      # Q2: on-axis, axis-aligned -> sigma_xy == 0 -> skipped
    add((0.3, -0.2, 2.5), (-1.5, -2.5, -2.0), rot=(0.9, 0.1, 0.3, -0.2))
    add((0.1, 0.1, -1.0), (-2.0, -2.0, -2.0), rot=(0.7, 0.2, 0.1, 0.1))   # behind the camera
    add((0.05, 0.02, 0.15), (-3.0, -3.0, -3.0), rot=(0.7, 0.2, 0.1, 0.1))  # z < 0.2 cull
    add((0.05, 0.02, 0.21), (-4.0, -4.0, -4.0), rot=(0.7, -0.2, 0.1, 0.3))  # just past the cull plane
    add((40.0, 0.0, 5.0), (-2.0, -2.0, -2.0), rot=(0.6, 0.2, 0.1, 0.5))    # far off-screen right
    add((0.0, -30.0, 4.0), (-2.0, -2.0, -2.0), rot=(0.6, 0.2, 0.1, 0.5))   # far off-screen top
    add((0.2, 0.1, 6.0), (1.0, 0.8, 0.5), rot=(0.8, 0.3, -0.2, 0.4), opacity=-1.0)  # covers the whole frame
    for k in range(12):                                           # opaque stack: hits the 0.99 alpha cap
        add((-0.4 + 0.01 * k, 0.3, 2.0 + 0.05 * k), (-1.2, -1.3, -1.4), rot=(0.9, 0.05 * k, 0.1, 0.2),
            opacity=12.0, dc=(1.5 - 0.2 * k, -0.5 + 0.2 * k, 0.3))
    add((1.9, 1.15, 3.3), (-1.0, -1.2, -1.1), rot=(0.5, 0.5, 0.2, 0.1))    # straddles the bottom-right corner
    add((-1.9, -1.2, 3.3), (-1.0, -1.2, -1.1), rot=(0.5, -0.5, 0.2, 0.1))  # straddles the top-left corner
    add((0.5, 0.4, 3.0), (-6.0, -6.0, -6.0), rot=(0.5, 0.1, 0.2, 0.1), opacity=-9.0)  # never reaches 1/255
    add((-0.6, 0.1, 2.8), (-0.5, -4.5, -4.5), rot=(0.9, 0.0, 0.0, 0.4))    # very anisotropic needle
    n = len(rows)
    rng = np.random.default_rng(33)
    xyz = np.array([r[0] for r in rows], np.float32)
    ls = np.array([r[1] for r in rows], np.float32)
    rot = np.array([r[2] for r in rows], np.float32)
    op = np.array([r[3] for r in rows], np.float32)
    dc = np.array([r[4] for r in rows], np.float32)
    rest = (rng.standard_normal((n, 45)) * 0.15).astype(np.float32)
    return syn._columns_from_blocks(xyz, ls, rot, op, dc, rest)


def make_f3(ref, out_dir):
    """edge cases; W,H not multiples of 16; scale-factor 4 (Q3); image_id != position (Q4)."""
    W, H = 150, 93
    hand = edge_columns()
    filler = fixture_scene(400, 77, 2.4)
    # filler sits around the origin; push it in front of the identity camera
    filler["z"] = (filler["z"] + np.float32(4.0)).astype(np.float32)
    cols = {k: np.concatenate([hand[k], filler[k]]) for k in hand}
    ident = syn.Pose(17, np.array([1.0, 0.0, 0.0, 0.0]), np.zeros(3), "edge_a.png")
    tilted = syn.look_at_pose((0.5, -3.0, 1.0), (0.0, 0.0, 3.0), 42, "edge_b.png")
    poses = [ident, tilted]
    fx = syn.pinhole_focal(W, 70.0)
    cases = {}
    # (a) identity camera, consistent scale-factor 2
    # (b) tilted camera, scale-factor 4 run over the same image size: focal stays fx_full/2 (Q3)
    for tag, pose, sf in (("a", ident, 2), ("b", tilted, 4)):
        r, dt = run_reference(ref, cols, poses, pose.image_id, W, H, 2 * fx, 2.1 * fx, 2 * W, 2 * H, scale_factor=sf)
        mid = pack_intermediates(r)
        cases[tag] = (pose, sf, r, mid)
        print(f"f3{tag}: drawn {len(r.draw_order)}/{len(cols['x'])} in {dt:.1f}s")
    d = {f"ply_{k}": v for k, v in cols.items()}
    d.update(width=np.int64(W), height=np.int64(H), fx_full=np.float64(2 * fx), fy_full=np.float64(2.1 * fx),
             cam_width=np.int64(2 * W), cam_height=np.int64(2 * H), n_hand=np.int64(len(hand["x"])))
    for tag, (pose, sf, r, mid) in cases.items():
        d[f"{tag}_qvec"], d[f"{tag}_tvec"] = np.asarray(pose.qvec), np.asarray(pose.tvec)
        d[f"{tag}_image_id"], d[f"{tag}_scale_factor"] = np.int64(pose.image_id), np.int64(sf)
        d[f"{tag}_image"] = r.frames[0].astype(np.float32)
        for k in ("rgb", "screen_means", "tile_bboxes", "pixel_bboxes", "sigmas", "opacity", "draw_order",
                  "cam_means", "cov2d", "cov3d"):
            d[f"{tag}_{k}"] = mid[k]
    # Q4: a scene_index that is not a COLMAP image_id must raise KeyError
    try:
        run_reference(ref, cols, poses, 0, W, H, 2 * fx, 2.1 * fx, 2 * W, 2 * H)
        d["q4_keyerror"] = np.bool_(False)
    except KeyError:
        d["q4_keyerror"] = np.bool_(True)
    np.savez_compressed(os.path.join(out_dir, "f3_edge.npz"), **d)
    print(f"f3: q4 KeyError on missing image_id: {bool(d['q4_keyerror'])}")


def make_f4(ref, out_dir):
    """medium: N=20000 at 960x540.  Inputs are regenerated from the seed by the test, so only the
    recipe, a 256x256 crop, float64 row/column sums and a 4x-decimated frame are stored."""
    W, H = 960, 540
    n, seed, shift = 20000, 360, 0.9
    cols = fixture_scene(n, seed, shift)
    poses = syn.ring_cameras(25)
    fx = syn.pinhole_focal(W)
    pose = poses[3]
    r, dt = run_reference(ref, cols, poses, pose.image_id, W, H, 2 * fx, 2 * fx, 2 * W, 2 * H)
    img = r.frames[0].astype(np.float32)
    d = dict(
        n=np.int64(n), seed=np.int64(seed), scale_shift=np.float64(shift), pose_index=np.int64(3),
        qvec=np.asarray(pose.qvec), tvec=np.asarray(pose.tvec), width=np.int64(W), height=np.int64(H),
        fx_full=np.float64(2 * fx), fy_full=np.float64(2 * fx), cam_width=np.int64(2 * W), cam_height=np.int64(2 * H),
        crop_y0=np.int64(142), crop_x0=np.int64(352),
        crop=img[142:398, 352:608].copy(),
        row_sums=img.astype(np.float64).sum(axis=(1, 2)), col_sums=img.astype(np.float64).sum(axis=(0, 2)),
        decimated=img[::4, ::4].copy(), n_drawn=np.int64(len(r.draw_order)),
        ply_checksum=np.float64(sum(float(np.asarray(v, np.float64).sum()) for v in cols.values())),
        reference_seconds=np.float64(dt),
    )
    np.savez_compressed(os.path.join(out_dir, "f4_medium.npz"), **d)
    print(f"f4: drawn {len(r.draw_order)}/{n} in {dt:.1f}s (reference end-to-end), mean {img.mean():.4f}")


def make_f5(ref, out_dir):
    """deep stacks: the fuzz case that needed the campaign's looser per-pixel threshold (tools/fuzz_parity.py, case-seed
    2519059510838425248: 45 918 gaussians with scales blown up e-fold on a 15x360 frame, hundreds of semi-transparent layers
    per pixel).  The scene is rebuilt from the seed by fuzz_parity.build_case; only the reference's frame is stored."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import fuzz_parity

    case_seed, max_n = 2519059510838425248, 60000
    c = fuzz_parity.build_case(case_seed, max_n)
    W, H, pose = c["W"], c["H"], c["pose"]
    _, _, fx_full, fy_full, cam_w, cam_h, _, _ = c["args"]
    cols = c["cols"]
    r, dt = run_reference(ref, cols, [pose], pose.image_id, W, H, fx_full, fy_full, int(cam_w), int(cam_h))
    img = r.frames[0].astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "f5_deep_stack.npz"), case_seed=np.int64(case_seed), max_n=np.int64(max_n), n=np.int64(c["n"]),
                        width=np.int64(W), height=np.int64(H), image=img, n_drawn=np.int64(len(r.draw_order)),
                        reference_seconds=np.float64(dt))
    print(f"f5: n {c['n']} {W}x{H}, drawn {len(r.draw_order)} in {dt:.1f}s, mean {img.mean():.4f}, max {img.max():.4f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="f1,f2,f3,f4")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    ref = import_reference()
    for tag in args.only.split(","):
        {"f1": make_f1, "f2": make_f2, "f3": make_f3, "f4": make_f4, "f5": make_f5}[tag](ref, args.out)


if __name__ == "__main__":
    main()
