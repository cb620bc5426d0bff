"""ctypes binding of csrc/libgsr.so (C ABI: include/gsr.h).

There is NO fallback: if the HIP library is missing or does not load, importing this
module raises.  Nothing under oracle/ is ever imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: its wheel bundles the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  libgsr.so
# must bind to that same runtime, or the streams / device pointers torch hands us belong to another one
# ("no ROCm-capable device is detected").  Loading torch before the dlopen below makes the SONAME resolve to it.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_HERE, "csrc", "libgsr.so")  # the override is for A/B tooling only

GSR_OK = 0
GSR_ERR_BAD_ARG = -1
GSR_ERR_WORKSPACE = -2
GSR_ERR_PAIR_OVERFLOW = -3
GSR_ERR_HIP = -4
GSR_ERR_SORT_PASSES = -5
GSR_MAX_PAIRS = 0xFFFFE000  # include/gsr.h
GSR_MAX_BATCH_VIEWS = 8
GSR_BOUNDS_BLOCK = 64


class GsrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libgsr error {code}: {message}")
        self.code = code


class GsrPairOverflow(GsrError):
    pass


class GsrSortPasses(GsrError):
    """The frame's depth keys needed more radix passes than GsrOptions.depth_sort_passes enqueued: the frame is wrong."""


class GsrScene(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("means", C.c_void_p),
        ("log_scales", C.c_void_p),
        ("quats", C.c_void_p),
        ("opacity_logit", C.c_void_p),
        ("sh", C.c_void_p),
        ("sh_degree", C.c_int32),
        ("sh_dtype", C.c_int32),
        ("block_bounds", C.c_void_p),
    ]


class GsrCamera(C.Structure):
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


class GsrOptions(C.Structure):
    _fields_ = [
        ("reference_compat", C.c_int32),
        ("early_out_T", C.c_float),
        ("tile_row_begin", C.c_int32),
        ("tile_row_step", C.c_int32),
        ("output_layout", C.c_int32),
        ("no_footprint_cull", C.c_int32),
        ("blend_impl", C.c_int32),
        ("draw_limit", C.c_int32),
        ("output_dtype", C.c_int32),
        ("depth_sort_passes", C.c_int32),
        ("accum_dtype", C.c_int32),
        ("keep_flags", C.c_int32),
        ("saturation_rule", C.c_int32),
        ("fine_binning", C.c_int32),
        ("shard_preprocess", C.c_int32),
        ("blend_pipe_tiles", C.c_int32),
        ("no_order_hint", C.c_int32),
        ("colour_stage", C.c_int32),
        ("sh_dense_min", C.c_int32),
        ("batch_views", C.c_int32),
        ("tile_row_block", C.c_int32),
    ]


class GsrStats(C.Structure):
    _fields_ = [
        ("n_visible", C.c_uint32),
        ("n_pairs_bbox", C.c_uint32),
        ("n_pairs", C.c_uint32),
        ("overflow", C.c_uint32),
        ("max_list_len", C.c_uint32),
        ("sort_passes", C.c_uint32),
        ("wave_entries", C.c_uint64),
        ("fetched_entries", C.c_uint64),
        ("colour_evals", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if not k.startswith("_")}


class GsrDebugOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("cov3d", "cam_means", "cov2d", "screen_means", "tile_bboxes", "sigmas", "pixel_bboxes", "rgb", "opacity")]


EXPORTS = [
    "gsr_version", "gsr_last_error", "gsr_default_options", "gsr_camera_setup", "gsr_workspace_bytes",
    "gsr_preprocess", "gsr_bin_sort", "gsr_blend", "gsr_render_forward", "gsr_read_stats", "gsr_sh_to_rgb", "gsr_cov3d",
    "gsr_render_batch", "gsr_render_batch_slots", "gsr_scene_order", "gsr_scene_order_bytes", "gsr_scene_bounds", "gsr_block_visibility", "gsr_project_to_camera_space", "gsr_compute_2d_covariance", "gsr_compute_covering_bbox", "gsr_rasterize_gaussian",
]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C torch-gaussian-splatting-rasterizer_amd/csrc).  There is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
    dp = C.POINTER(C.c_double)
    L.gsr_version.restype = C.c_int
    L.gsr_last_error.restype = C.c_char_p
    L.gsr_default_options.argtypes = [C.POINTER(GsrOptions)]
    L.gsr_default_options.restype = None
    L.gsr_camera_setup.argtypes = [dp, dp, C.c_double, C.c_double, i64, i64, i32, i32, C.POINTER(GsrCamera)]
    L.gsr_workspace_bytes.argtypes = [i64, i32, i32, i64, C.POINTER(sz)]
    L.gsr_preprocess.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrOptions), vp, sz,
                                 C.POINTER(GsrDebugOut), vp]
    L.gsr_bin_sort.argtypes = [i64, C.POINTER(GsrCamera), C.POINTER(GsrOptions), i64, vp, sz, vp]
    L.gsr_blend.argtypes = [C.POINTER(GsrScene), i64, C.POINTER(GsrCamera), C.POINTER(GsrOptions), i64, vp, sz, vp, vp, vp]
    L.gsr_render_forward.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrOptions), i64, vp, sz, vp, vp, vp]
    L.gsr_render_batch.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrCamera), i32, C.POINTER(GsrOptions), i64, vp, sz, vp, i64, vp]
    L.gsr_render_batch_slots.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrCamera), i32, C.POINTER(GsrOptions), i64, C.POINTER(vp), sz,
                                         C.POINTER(vp), i32, vp, i64]
    L.gsr_read_stats.argtypes = [vp, sz, C.POINTER(GsrStats), vp]
    L.gsr_scene_order_bytes.argtypes = [i64, C.POINTER(sz)]
    L.gsr_scene_order.argtypes = [i64, vp, vp, vp, sz, vp]
    L.gsr_scene_bounds.argtypes = [i64, vp, vp, vp, vp]
    L.gsr_block_visibility.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrOptions), vp, vp]
    L.gsr_sh_to_rgb.argtypes = [i64, vp, vp, C.POINTER(C.c_float), i32, vp, vp]
    L.gsr_cov3d.argtypes = [i64, vp, vp, vp, vp]
    f16 = C.POINTER(C.c_float)
    L.gsr_project_to_camera_space.argtypes = [i64, vp, f16, vp, vp]
    L.gsr_compute_2d_covariance.argtypes = [i64, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, f16, vp, vp]
    L.gsr_compute_covering_bbox.argtypes = [i64, vp, vp, C.c_double, C.c_double, vp, vp]
    L.gsr_rasterize_gaussian.argtypes = [i64, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    for name in EXPORTS:
        if name not in ("gsr_last_error", "gsr_default_options"):
            getattr(L, name).restype = C.c_int
    return L


lib = _load()


def check(rc: int) -> None:
    if rc == GSR_OK:
        return
    msg = (lib.gsr_last_error() or b"").decode("utf-8", "replace")
    raise {GSR_ERR_PAIR_OVERFLOW: GsrPairOverflow, GSR_ERR_SORT_PASSES: GsrSortPasses}.get(rc, GsrError)(rc, msg)


def default_options() -> GsrOptions:
    o = GsrOptions()
    lib.gsr_default_options(C.byref(o))
    return o


def camera_setup(qvec, tvec, fx_full, fy_full, cam_width, cam_height, width, height) -> GsrCamera:
    q = (C.c_double * 4)(*[float(v) for v in qvec])
    t = (C.c_double * 3)(*[float(v) for v in tvec])
    cam = GsrCamera()
    check(lib.gsr_camera_setup(q, t, float(fx_full), float(fy_full), int(cam_width), int(cam_height), int(width),
                               int(height), C.byref(cam)))
    return cam


def workspace_bytes(n: int, width: int, height: int, max_pairs: int) -> int:
    out = C.c_size_t(0)
    check(lib.gsr_workspace_bytes(int(n), int(width), int(height), int(max_pairs), C.byref(out)))
    return int(out.value)
