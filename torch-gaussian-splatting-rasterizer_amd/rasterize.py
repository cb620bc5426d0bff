"""Drop-in for the reference's `rasterize.py`: the same constants, helper names and `run_rasterization`
command, with the per-gaussian work done by libgsr (HIP, gfx950) instead of torch ops and a Python loop.

Reference lines each piece stands in for are cited per function.  New here: `render_scene`, which returns the
frame (the reference's command only shows a matplotlib figure and returns None).
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os
from typing import Optional, Tuple

import click
import numpy as np
import torch

from . import renderer
from ._lib import check, lib
from .ply import PlyData, PlyElement
from .spherical_harmonics import sh_to_rgb
from .utils import pack_gaussians, read_color_components, read_scene

logger = logging.getLogger(__name__)

# reference rasterize.py:29-38
Z_FAR = 100.0
Z_NEAR = 0.01
GAUSSIAN_SPREAD = 3
BLOCK_SIZE = 16
MAX_GAUSSIAN_DENSITY = 0.99
MIN_ALPHA = 1 / 255


def _gpu(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor: libgsr has no CPU path")
    return t


def _stream(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _mat16(m: torch.Tensor):
    flat = m.detach().float().cpu().contiguous().view(-1).tolist()
    if len(flat) != 16:
        raise ValueError("expected a 4x4 matrix")
    return (C.c_float * 16)(*flat)


def quaternion_to_rotation_matrix(quaternion: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) stacked as [4,N] -> rotation matrices [3,3,N] float32 (reference :41-56; no normalisation inside)."""
    w, x, y, z = quaternion[0], quaternion[1], quaternion[2], quaternion[3]
    rows = (
        (1 - 2 * y ** 2 - 2 * z ** 2, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w),
        (2 * x * y + 2 * z * w, 1 - 2 * x ** 2 - 2 * z ** 2, 2 * y * z - 2 * x * w),
        (2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x ** 2 - 2 * y ** 2),
    )
    return torch.stack([torch.stack(r) for r in rows]).float()


def get_world_to_camera_matrix(normalized_qvec: torch.Tensor, tvec: torch.Tensor) -> torch.Tensor:
    """4x4 [[R, t],[0, 1]] in float32 from a COLMAP pose (reference :59-77; translation is +tvec)."""
    m = torch.zeros((4, 4))
    m[:3, :3] = quaternion_to_rotation_matrix(normalized_qvec.unsqueeze(1)).squeeze(-1)
    m[:3, 3] = tvec
    m[3, 3] = 1
    return m


def get_projection_matrix(fov_x: float, fov_y: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign = +1 (reference :123-151), float64 arithmetic stored as float32."""
    tx, ty = math.tan(fov_x / 2), math.tan(fov_y / 2)
    top, right = ty * Z_NEAR, tx * Z_NEAR
    bottom, left = -top, -right
    p = torch.zeros(4, 4)
    p[0, 0] = 2.0 * Z_NEAR / (right - left)
    p[1, 1] = 2.0 * Z_NEAR / (top - bottom)
    p[0, 2] = (right + left) / (right - left)
    p[1, 2] = (top + bottom) / (top - bottom)
    p[3, 2] = 1.0
    p[2, 2] = Z_FAR / (Z_FAR - Z_NEAR)
    p[2, 3] = -(Z_FAR * Z_NEAR) / (Z_FAR - Z_NEAR)
    return p


def project_to_camera_space(gaussian_means: torch.Tensor, world_to_camera: torch.Tensor) -> torch.Tensor:
    """means @ w2c[:3,:3] + w2c[3,:3] with w2c in the reference's transposed (row-vector) form (reference :80-86)."""
    means = _gpu(gaussian_means, "gaussian_means").contiguous().float()
    out = torch.empty_like(means)
    check(lib.gsr_project_to_camera_space(means.shape[0], means.data_ptr(), _mat16(world_to_camera), out.data_ptr(),
                                          _stream(means.device)))
    return out


def get_covariance_matrix_from_mesh(mesh, device="cuda") -> torch.Tensor:
    """3D covariances R S S^T R^T of every gaussian in a ply, [N,3,3] float32 on `device` (reference :89-120)."""
    el = mesh.elements[0]
    log_scales = torch.from_numpy(np.stack([np.asarray(el[f"scale_{i}"], np.float32) for i in range(3)], 1)).to(device)
    quats = torch.from_numpy(np.stack([np.asarray(el[f"rot_{i}"], np.float32) for i in range(4)], 1)).to(device)
    _gpu(log_scales, "device")
    out = torch.empty((log_scales.shape[0], 3, 3), dtype=torch.float32, device=log_scales.device)
    check(lib.gsr_cov3d(log_scales.shape[0], log_scales.contiguous().data_ptr(), quats.contiguous().data_ptr(), out.data_ptr(),
                        _stream(out.device)))
    return out


def compute_covering_bbox(screen_means: torch.Tensor, projected_covariances: torch.Tensor, width: float, height: float) -> torch.Tensor:
    """Tile-unit bounding boxes [N,4] int64 from 3-sigma radii (reference :154-198)."""
    sm = _gpu(screen_means, "screen_means").contiguous().float()
    cov = projected_covariances.to(sm.device).contiguous().float().view(-1, 4)
    out = torch.empty((sm.shape[0], 4), dtype=torch.int64, device=sm.device)
    check(lib.gsr_compute_covering_bbox(sm.shape[0], sm.data_ptr(), cov.data_ptr(), float(width), float(height), out.data_ptr(),
                                        _stream(sm.device)))
    return out


def compute_2d_covariance(cov_matrices, camera_space_points, tan_fov_x, tan_fov_y, focals, world_to_camera) -> torch.Tensor:
    """EWA projection of the 3D covariances to screen space, [N,2,2] incl. the 0.3 low-pass (reference :201-252).
    `focals` are the full-resolution fx, fy; they are halved inside exactly like the reference (quirk Q3)."""
    cov3 = _gpu(cov_matrices, "cov_matrices").contiguous().float().view(-1, 9)
    cam = camera_space_points.to(cov3.device).contiguous().float()
    out = torch.empty((cov3.shape[0], 2, 2), dtype=torch.float32, device=cov3.device)
    check(lib.gsr_compute_2d_covariance(cov3.shape[0], cov3.data_ptr(), cam.data_ptr(), float(tan_fov_x), float(tan_fov_y),
                                        float(focals[0]), float(focals[1]), _mat16(world_to_camera), out.data_ptr(),
                                        _stream(cov3.device)))
    return out


def rasterize_gaussian(gaussian_index: int, bboxes: torch.Tensor, screen: torch.Tensor, screen_means: torch.Tensor,
                       sigmas: torch.Tensor, rgb: torch.Tensor, opacity_buffer: torch.Tensor,
                       opacity: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Blend ONE gaussian over its pixel rect, in place: screen [W,H,3], opacity_buffer [W,H] (reference :255-305)."""
    _gpu(screen, "screen")
    for t, name, dt in ((bboxes, "bboxes", torch.int64), (screen, "screen", torch.float32), (screen_means, "screen_means", torch.float32),
                        (sigmas, "sigmas", torch.float32), (rgb, "rgb", torch.float32), (opacity_buffer, "opacity_buffer", torch.float32),
                        (opacity, "opacity", torch.float32)):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == dt):
            raise ValueError(f"{name} must be a contiguous {dt} CUDA tensor")
    w, h = screen.shape[0], screen.shape[1]
    check(lib.gsr_rasterize_gaussian(int(gaussian_index), bboxes.shape[0], bboxes.data_ptr(), screen.data_ptr(), screen_means.data_ptr(),
                                     sigmas.data_ptr(), rgb.data_ptr(), opacity_buffer.data_ptr(), opacity.data_ptr(), w, h,
                                     _stream(screen.device)))
    return screen, opacity_buffer


# ------------------------------------------------------------------------------------------------------
def load_view(input_dir: str, scene_index: int, scale_factor: int):
    """Everything the reference reads before touching the ply (reference :328-345): the COLMAP pose keyed by
    image_id (Q4), camera_id 1 intrinsics, and the frame size taken from images_{scale_factor}/<name>."""
    from PIL import Image

    scenes, cam_info = read_scene(path_to_scene=input_dir)
    scene = scenes[scene_index]  # KeyError if no image has this id, like the reference
    gt_img_path = os.path.join(input_dir, f"images_{scale_factor}", scene.name)
    with Image.open(gt_img_path) as img:
        width, height = img.size
    cam0 = cam_info[1]
    fx, fy = float(cam0.params[0]), float(cam0.params[1])
    cam = renderer.make_camera(scene.qvec, scene.tvec, fx, fy, int(cam0.width), int(cam0.height), int(width), int(height))
    return cam, gt_img_path


def render_scene(input_dir: str, trained_model_path: str, scene_index: int = 0, scale_factor: int = 2, device: str = "cuda",
                 reference_compat: bool = True, early_out_T: float = 0.0, scene_order: str = "morton") -> torch.Tensor:
    """The render call of the reference (:327-446) as a function: returns the frame [H,W,3] float32 on `device`.
    scene_order: "morton" (default) uploads the gaussians along a Morton curve of their means (~10 % faster frames), "file" keeps the
    .ply's order.  Same frame either way except where gaussians at EXACTLY equal depth overlap: those are drawn in storage order
    (the reference's torch.sort, :425, leaves them undefined; "file" is what its CPU sort does in practice)."""
    if scene_order not in ("morton", "file"):
        raise ValueError("scene_order: 'morton' or 'file'")
    cam, _ = load_view(input_dir, scene_index, scale_factor)
    ply_path = os.path.join(trained_model_path, "point_cloud/iteration_30000/point_cloud.ply")
    logger.info("Fetching trained model from: %s", ply_path)
    scene = renderer.GaussianScene.from_ply(ply_path, device=device, spatial_order=scene_order == "morton")
    return renderer.Rasterizer(scene).render(cam, renderer.make_options(reference_compat=reference_compat, early_out_T=early_out_T))


def _to_png_array(frame: torch.Tensor) -> np.ndarray:
    """The reference's frame dump: (screen.transpose(1,0) * 255).astype(uint8), truncating (rasterize.py:449)."""
    return (frame.cpu().numpy() * 255.0).astype(np.uint8)


def render_progressive(input_dir: str, trained_model_path: str, output_path: str, scene_index: int = 0, scale_factor: int = 2,
                       every: int = 1000, framerate: int = 20, device: str = "cuda", scene_order: str = "morton") -> torch.Tensor:
    """--generate_video (reference :427-429,:448-466): a PNG after gaussian number 1, 1001, 2001, ... of the draw
    order (the reference saves when iteration_step % 1000 == 0, right after drawing), 2 s of padding frames that
    repeat the LAST SAVED frame (the reference re-saves its stale `img`), then ffmpeg if it is installed.
    Returns the final frame."""
    import shutil
    import subprocess

    from PIL import Image

    cam, _ = load_view(input_dir, scene_index, scale_factor)
    ply_path = os.path.join(trained_model_path, "point_cloud/iteration_30000/point_cloud.ply")
    R = renderer.Rasterizer(renderer.GaussianScene.from_ply(ply_path, device=device, spatial_order=scene_order == "morton"))
    final = R.render(cam, renderer.make_options(draw_limit=2 ** 31 - 1))  # draw_limit > 0: counters follow the reference's order
    n_drawn = int(R.last_stats["n_visible"])
    img_dir = os.path.join(output_path, "images")
    os.makedirs(img_dir, exist_ok=True)
    img = None
    for step in range(0, n_drawn, every):
        frame = R.render(cam, renderer.make_options(draw_limit=step + 1))
        img = Image.fromarray(_to_png_array(frame))
        img.save(os.path.join(img_dir, f"image_iter_{str(step).zfill(7)}.png"))
    if img is not None:
        for i in range(1, 2 * framerate + 1):
            img.save(os.path.join(img_dir, f"image_iter_{str(n_drawn + 1000 * i + 1).zfill(7)}.png"))
    width, height = cam.width, cam.height
    video_path = os.path.join(output_path, "video_render.mp4")
    if shutil.which("ffmpeg"):
        if os.path.exists(video_path):
            os.remove(video_path)
        pattern = os.path.join(img_dir, "image_iter_*.png")
        subprocess.run(["ffmpeg", "-framerate", str(framerate), "-pattern_type", "glob", "-i", pattern, "-r", "10", "-vcodec", "libx264",
                        "-s", f"{width - (width % 2)}x{height - (height % 2)}", "-pix_fmt", "yuv420p", video_path], check=True)
    else:
        logger.warning("ffmpeg not found: wrote the frames to %s, no video", img_dir)
    return final


@click.command()
@click.option("--input_dir", type=str, default="")
@click.option("--trained_model_path", type=str, default="")
@click.option("--output_path", type=str, default="")
@click.option("--scene-index", type=int, default=0)
@click.option("--scale-factor", type=int, default=2)
@click.option("--generate_video", is_flag=True, type=bool, default=False)
@click.option("--scene-order", type=click.Choice(["morton", "file"]), default="morton",
              help="storage order of the gaussians in HBM: along a Morton curve of their means (default, ~10 % faster) or as in the "
                   ".ply.  Same frame except where gaussians at EXACTLY equal depth overlap (drawn in storage order; the reference "
                   "leaves their order undefined, 'file' is what its CPU sort does in practice).")
def run_rasterization(input_dir: str, trained_model_path, output_path: Optional[str], scene_index: int = 0,
                      scale_factor: int = 2, generate_video: bool = False, scene_order: str = "morton") -> None:
    """Same six options as the reference's command (:308-314) + --scene-order.  Instead of a matplotlib window the frame is
    written to <output_path>/render.npy and render.png (uint8 truncation like the reference's frame dumps, :449)."""
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    logger.info("Fetching scenes from: %s", input_dir)
    if generate_video:
        image = render_progressive(input_dir, trained_model_path, output_path, scene_index, scale_factor, scene_order=scene_order)
    else:
        image = render_scene(input_dir, trained_model_path, scene_index, scale_factor, scene_order=scene_order)
    if output_path:
        from PIL import Image

        os.makedirs(output_path, exist_ok=True)
        arr = image.cpu().numpy()
        np.save(os.path.join(output_path, "render.npy"), arr)
        Image.fromarray((arr * 255.0).astype(np.uint8)).save(os.path.join(output_path, "render.png"))
    return None


if __name__ == "__main__":
    run_rasterization()
