"""Degree-0..3 real spherical harmonics -> RGB.  Same surface as the reference's `spherical_harmonics.py`
(constants at :4-24, `sh_to_rgb` at :27-73); the evaluation runs in libgsr's `gsr_sh_to_rgb` kernel."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib

# Real SH normalisation constants in the INRIA sign convention (reference spherical_harmonics.py:4-24)
SH_0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def camera_center(world_view_transform: torch.Tensor) -> torch.Tensor:
    """inverse(world_view)[3,:3] (reference :35) for the row-vector 4x4 [[A,0],[t,1]]: -t @ inverse(A), in float64."""
    m = world_view_transform.detach().double().cpu()
    return (-(m[3, :3] @ torch.linalg.inv(m[:3, :3]))).float()


def sh_to_rgb(xyz: torch.Tensor, sh: torch.Tensor, world_view_transform: torch.Tensor, degree: int = 0) -> torch.Tensor:
    """View-dependent colour of every gaussian: [N,3] means, [N,16,3] coefficients -> [N,3] in [0,1].

    Direction = normalised (mean - camera centre); `+0.5` offset and clamp to [0,1] on both sides as the
    reference does (:69-71)."""
    if not xyz.is_cuda:
        raise RuntimeError("sh_to_rgb: tensors must be on the GPU (libgsr has no CPU path)")
    if not 0 <= int(degree) <= 3:
        raise ValueError("degree must be 0..3")
    xyz = xyz.contiguous().float()
    sh = sh.to(xyz.device).contiguous().float()
    n = xyz.shape[0]
    if tuple(sh.shape) != (n, 16, 3):
        raise ValueError(f"sh must be [N,16,3], got {tuple(sh.shape)}")
    cc = (C.c_float * 3)(*camera_center(world_view_transform).tolist())
    out = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    check(lib.gsr_sh_to_rgb(n, xyz.data_ptr(), sh.data_ptr(), cc, int(degree), out.data_ptr(),
                            int(torch.cuda.current_stream(xyz.device).cuda_stream)))
    return out
