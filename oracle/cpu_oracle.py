"""ctypes front-end of the CPU ORACLE (oracle/gsr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this;
the product package never does.  See gsr_oracle.c for the reference citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgsr_oracle.so")


class OracleCamera(C.Structure):
    _fields_ = [
        ("w2c", C.c_float * 16),
        ("full_proj", C.c_float * 16),
        ("cam_center", C.c_float * 3),
        ("focal_x", C.c_float),
        ("focal_y", C.c_float),
        ("lim_x", C.c_float),
        ("lim_y", C.c_float),
        ("tan_fov_x", C.c_float),
        ("tan_fov_y", C.c_float),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.gsr_oracle_camera.argtypes = [dp, dp, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                        C.POINTER(OracleCamera)]
        L.gsr_oracle_camera.restype = None
        L.gsr_oracle_preprocess.argtypes = [C.c_int64, fp, fp, fp, fp, fp, C.c_int, C.POINTER(OracleCamera),
                                            fp, fp, fp, fp, ip, fp, ip, fp, fp]
        L.gsr_oracle_preprocess.restype = None
        L.gsr_oracle_depth_order.argtypes = [C.c_int64, fp, ip]
        L.gsr_oracle_depth_order.restype = None
        L.gsr_oracle_composite.argtypes = [C.c_int64, ip, ip, fp, fp, fp, fp, C.c_int32, C.c_int32, C.c_int64, C.c_int,
                                           fp, fp]
        L.gsr_oracle_composite.restype = C.c_int64
        L.gsr_oracle_render.argtypes = [C.c_int64, fp, fp, fp, fp, fp, C.c_int, C.POINTER(OracleCamera), C.c_int, fp, fp]
        L.gsr_oracle_render.restype = C.c_int64
        L.gsr_oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def camera(qvec, tvec, fx_full: float, fy_full: float, cam_width: int, cam_height: int, width: int, height: int) -> OracleCamera:
    cam = OracleCamera()
    q = np.ascontiguousarray(qvec, np.float64)
    t = np.ascontiguousarray(tvec, np.float64)
    dp = C.POINTER(C.c_double)
    lib().gsr_oracle_camera(q.ctypes.data_as(dp), t.ctypes.data_as(dp), float(fx_full), float(fy_full),
                            int(cam_width), int(cam_height), int(width), int(height), C.byref(cam))
    return cam


def preprocess(packed: Dict[str, np.ndarray], cam: OracleCamera, sh_degree: int = 3) -> Dict[str, np.ndarray]:
    n = packed["means"].shape[0]
    out = dict(
        cov3d=np.empty((n, 3, 3), np.float32), cam_means=np.empty((n, 3), np.float32),
        cov2d=np.empty((n, 2, 2), np.float32), screen_means=np.empty((n, 2), np.float32),
        tile_bboxes=np.empty((n, 4), np.int64), sigmas=np.empty((n, 3), np.float32),
        pixel_bboxes=np.empty((n, 4), np.int64), rgb=np.empty((n, 3), np.float32), opacity=np.empty((n,), np.float32),
    )
    lib().gsr_oracle_preprocess(
        n, _f(packed["means"]), _f(packed["log_scales"]), _f(packed["quats"]), _f(packed["opacity_logit"]),
        _f(packed["sh"]), sh_degree, C.byref(cam),
        _f(out["cov3d"]), _f(out["cam_means"]), _f(out["cov2d"]), _f(out["screen_means"]), _i(out["tile_bboxes"]),
        _f(out["sigmas"]), _i(out["pixel_bboxes"]), _f(out["rgb"]), _f(out["opacity"]),
    )
    return out


def depth_order(cam_means: np.ndarray) -> np.ndarray:
    order = np.empty((cam_means.shape[0],), np.int64)
    lib().gsr_oracle_depth_order(cam_means.shape[0], _f(np.ascontiguousarray(cam_means)), _i(order))
    return order


def composite(order, pre: Dict[str, np.ndarray], width: int, height: int, limit: int = -1, threads: int = 1):
    """The per-gaussian loop.  Returns (screen [W,H,3], transmittance [W,H], n_drawn) in the reference's layout."""
    screen = np.zeros((width, height, 3), np.float32)
    trans = np.ones((width, height), np.float32)
    drawn = lib().gsr_oracle_composite(
        len(order), _i(np.ascontiguousarray(order)), _i(pre["pixel_bboxes"]), _f(pre["screen_means"]), _f(pre["sigmas"]),
        _f(pre["rgb"]), _f(pre["opacity"]), width, height, limit, threads, _f(screen), _f(trans))
    return screen, trans, int(drawn)


def render(packed: Dict[str, np.ndarray], cam: OracleCamera, sh_degree: int = 3, threads: Optional[int] = None,
           want_T: bool = False):
    """Whole frame -> image [H,W,3] float32 (= screen.transpose(1,0), rasterize.py:471)."""
    n = packed["means"].shape[0]
    img = np.empty((cam.height, cam.width, 3), np.float32)
    fT = np.empty((cam.height, cam.width), np.float32) if want_T else None
    if threads is None:
        threads = max_threads()
    drawn = lib().gsr_oracle_render(
        n, _f(packed["means"]), _f(packed["log_scales"]), _f(packed["quats"]), _f(packed["opacity_logit"]),
        _f(packed["sh"]), sh_degree, C.byref(cam), int(threads), _f(img), _f(fT) if want_T else None)
    if drawn < 0:
        raise MemoryError("oracle allocation failed")
    return (img, fT, int(drawn)) if want_T else (img, int(drawn))


def max_threads() -> int:
    return int(lib().gsr_oracle_max_threads())
