"""Host side of the render call: device-resident scene, camera block, workspace, frame launch.

This is the function the reference never had — its render call is the body of
`run_rasterization` (reference rasterize.py:347-446) and returns nothing; `Rasterizer.render`
returns the frame.  PyTorch only provides device memory and the stream; all per-gaussian and
per-pixel work happens in libgsr.so through the C ABI (include/gsr.h).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from ._lib import GsrCamera, GsrDebugOut, GsrOptions, GsrScene, GsrStats, check, lib
from .utils import pack_gaussians

TILE = 16
MAX_RETRIES = 6  # re-renders of one frame / batch after an exceeded bound (pair buffer, depth-sort passes) before giving up


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: libgsr has no CPU path")


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def morton_order(means: np.ndarray) -> np.ndarray:
    """Permutation that lays gaussians out along a Morton (Z-order) curve of their means: every axis rank-quantised to 10 bits (so
    the curve is balanced whatever the scene's extent), bits interleaved, stable.  A trained .ply is in no spatial order, and what
    a camera sees is a spatial region: in curve order the gaussians of a wave are neighbours, so waves are culled whole, the 192-B
    SH rows of the visible ones are contiguous (no partly used lines) and the blend's record gathers hit L2 more often."""
    n = len(means)
    if n == 0:
        return np.zeros(0, np.int64)

    def spread(v):  # 10 bits -> every third bit
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v

    q = [np.argsort(np.argsort(means[:, a], kind="stable"), kind="stable") * 1024 // n for a in range(3)]
    code = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2)
    return np.argsort(code, kind="stable")


def scene_order(means: torch.Tensor) -> torch.Tensor:
    """morton_order through the C ABI (gsr_scene_order: libgsr's own radix passes, ~2 ms at 6 M gaussians, no first-use kernel
    loading): int64 permutation on the device of `means`, element for element the numpy statement's."""
    _require_cuda(means, "means")
    n = int(means.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=means.device)
    m = means.contiguous().float()
    need = C.c_size_t(0)
    check(lib.gsr_scene_order_bytes(n, C.byref(need)))
    with torch.cuda.device(means.device):
        ws = torch.empty(int(need.value), dtype=torch.uint8, device=means.device)
        perm = torch.empty(n, dtype=torch.int32, device=means.device)  # uint32 on the wire; n < 2^31
        check(lib.gsr_scene_order(n, m.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(means.device)))
        return perm.to(torch.int64)


def morton_order_device(means: torch.Tensor) -> torch.Tensor:
    """morton_order with torch ops, on whatever device the means live on (CPU in the tests): the same permutation, element for
    element — per axis a stable argsort and its inverse give the ranks, the quantised ranks are interleaved into 30-bit codes, one
    more stable argsort orders them.  On the GPU the loaders use `scene_order` (the C ABI) instead: the first torch sort of a
    process costs 0.1-0.4 s of rocPRIM kernel loading."""
    n = int(means.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=means.device)

    def spread(v):  # 10 bits -> every third bit
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v

    ar = torch.arange(n, dtype=torch.int64, device=means.device)
    code = torch.zeros(n, dtype=torch.int64, device=means.device)
    for a in range(3):
        idx = torch.argsort(means[:, a].float(), stable=True)
        rank = torch.empty_like(idx)
        rank[idx] = ar
        code |= spread(rank * 1024 // n) << a
    return torch.argsort(code, stable=True)


class GaussianScene:
    """Camera-independent trained gaussians, resident in HBM in the layout of GsrScene.

    The loaders below upload the arrays along a Morton curve of the means (`spatial_order=True`, their default since round 4;
    `order` holds the permutation: scene index -> file index, None in file order).  A trained .ply is in no spatial order while a
    camera sees a spatial region: in curve order the preprocess culls whole waves, the SH rows of the visible gaussians are
    contiguous and the blend's record gathers hit L2 more often — ~10 % of the frame (bench.py's `file_order` leg).
    The frame is the same: the reference's depth sort orders the draw, not the storage order — except for gaussians at EXACTLY
    equal depth, whose mutual order the reference leaves undefined (torch.sort, rasterize.py:425, is unstable) and this library
    resolves by scene index.  Per-gaussian outputs (`Rasterizer.preprocess_debug`) come back in FILE order either way.
    The constructor takes device arrays as they are (no reordering): that is the C ABI's view."""

    FIELDS = ("means", "log_scales", "quats", "opacity_logit", "sh")

    def __init__(self, arrays: Mapping[str, torch.Tensor], sh_degree: int = 3, sh_half: bool = False):
        self.t: Dict[str, torch.Tensor] = {}
        self.bounds: Optional[torch.Tensor] = None   # GsrScene.block_bounds (build_bounds): [ceil(n / 256), 8] or None
        self.order_t: Optional[torch.Tensor] = None  # device, int64: scene index -> file index (None: file order)
        self._order_np: Optional[np.ndarray] = None
        self.order_ms = 0.0                          # what building the order and gathering the arrays cost at upload
        self.sh_half = bool(sh_half)
        for k in self.FIELDS:
            v = arrays[k]
            _require_cuda(v, k)
            self.t[k] = v.contiguous().half() if (k == "sh" and sh_half) else v.contiguous().float()
        self.n = int(self.t["means"].shape[0])
        self.device = self.t["means"].device
        self.sh_degree = int(sh_degree)
        shapes = {"means": (self.n, 3), "log_scales": (self.n, 3), "quats": (self.n, 4), "opacity_logit": (self.n,),
                  "sh": (self.n, 16, 3)}
        for k, shp in shapes.items():
            if tuple(self.t[k].shape) != shp:
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(self.t[k].shape)}")

    @property
    def order(self) -> Optional[np.ndarray]:
        """scene index -> file index as a numpy array (None: the scene is in file order)."""
        if self.order_t is None:
            return None
        if self._order_np is None:
            self._order_np = self.order_t.cpu().numpy()
        return self._order_np

    def sort_spatially(self) -> "GaussianScene":
        """Reorder the resident arrays along the Morton curve of the means (on the device; a scene already ordered is left alone)."""
        if self.order_t is not None or self.n == 0:
            return self
        with torch.cuda.device(self.device):  # the events (and the sorts' temporaries) belong to the scene's device, whichever is current
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            order = scene_order(self.t["means"])
            for k in self.FIELDS:
                self.t[k] = self.t[k].index_select(0, order).contiguous()
            t1.record()
            t1.synchronize()
        self.order_t, self._order_np, self.order_ms = order, None, float(t0.elapsed_time(t1))
        if self.bounds is not None:  # they described the old order
            self.build_bounds()
        return self

    def blocks_skipped(self, cam: GsrCamera, opts: Optional[GsrOptions] = None) -> torch.Tensor:
        """uint8 [blocks]: 1 where the preprocess skips the block for this view (gsr_block_visibility; needs build_bounds)."""
        if self.bounds is None:
            raise ValueError("the scene has no block bounds (build_bounds)")
        opts = opts or make_options()
        sc = self.c_struct()
        with torch.cuda.device(self.device):
            dead = torch.zeros(self.bounds.shape[0], dtype=torch.uint8, device=self.device)
            check(lib.gsr_block_visibility(C.byref(sc), C.byref(cam), C.byref(opts), dead.data_ptr(), _stream_ptr(self.device)))
        return dead

    def build_bounds(self) -> "GaussianScene":
        """Camera-independent block bounds of the arrays AS THEY LIE NOW (gsr_scene_bounds: per 256 consecutive gaussians the box of
        their means and their largest log-scale, 32 B per block): the preprocess then skips, unread, every block none of whose gaussians
        can be drawn in a view — bit-identical frames.  Pays off in spatial order (~40 % of a view's blocks); harmless in file order."""
        if self.n == 0:
            return self
        with torch.cuda.device(self.device):
            nblk = (self.n + _lib.GSR_BOUNDS_BLOCK - 1) // _lib.GSR_BOUNDS_BLOCK
            b = torch.empty((nblk, 8), dtype=torch.float32, device=self.device)
            check(lib.gsr_scene_bounds(self.n, self.t["means"].data_ptr(), self.t["log_scales"].data_ptr(), b.data_ptr(), _stream_ptr(self.device)))
        self.bounds = b
        return self

    @classmethod
    def from_columns(cls, columns, device="cuda", sh_degree: int = 3, sh_half: bool = False, spatial_order: bool = True) -> "GaussianScene":
        """`columns`: ply element / dict of float32 columns named as in the INRIA .ply."""
        return cls.from_packed(pack_gaussians(columns), device, sh_degree, sh_half, spatial_order)

    @classmethod
    def from_packed(cls, packed: Mapping[str, np.ndarray], device="cuda", sh_degree: int = 3, sh_half: bool = False,
                    spatial_order: bool = True) -> "GaussianScene":
        scene = cls({k: torch.from_numpy(np.ascontiguousarray(np.asarray(packed[k], np.float32))).to(device) for k in cls.FIELDS},
                    sh_degree, sh_half)
        # block bounds pay in spatial order (31 % of the bench frame's blocks are skipped unread); in file order no box is tight enough
        # to rule a block out and the flags kernel would run for nothing
        return scene.sort_spatially().build_bounds() if spatial_order else scene

    @classmethod
    def from_ply(cls, path: str, device="cuda", sh_degree: int = 3, sh_half: bool = False, spatial_order: bool = True) -> "GaussianScene":
        from .ply import PlyData

        return cls.from_columns(PlyData.read(path), device, sh_degree, sh_half, spatial_order)

    def c_struct(self) -> GsrScene:
        s = GsrScene()
        s.n = self.n
        for k in self.FIELDS:
            setattr(s, k, self.t[k].data_ptr())
        s.sh_degree = self.sh_degree
        s.sh_dtype = 1 if self.sh_half else 0
        s.block_bounds = self.bounds.data_ptr() if self.bounds is not None else None
        return s


def make_camera(qvec, tvec, fx_full: float, fy_full: float, cam_width: int, cam_height: int, width: int, height: int) -> GsrCamera:
    """COLMAP pose + full-res intrinsics + frame size -> GsrCamera (reference rasterize.py:336-345,:361-364)."""
    return _lib.camera_setup(qvec, tvec, fx_full, fy_full, cam_width, cam_height, width, height)


def make_options(reference_compat: bool = True, early_out_T: float = 0.0, tile_row_begin: int = 0, tile_row_step: int = 1,
                 output_layout: int = 0, no_footprint_cull: bool = False, blend_impl: int = 0, draw_limit: int = 0,
                 output_bf16: bool = False, depth_sort_passes: int = 0, keep_flags: bool = False, accum_bf16: bool = False,
                 saturation_rule: int = 0, fine_binning: bool = False, shard_preprocess: int = 0, blend_pipe_tiles: int = 0,
                 sh_dense_min: int = 0, colour_stage: int = 0, no_order_hint: bool = False, batch_views: int = 0,
                 tile_row_block: int = 0) -> GsrOptions:
    o = _lib.default_options()
    o.reference_compat = 1 if reference_compat else 0
    o.early_out_T = float(early_out_T)
    o.tile_row_begin = int(tile_row_begin)
    o.tile_row_step = int(tile_row_step)
    o.output_layout = int(output_layout)
    o.no_footprint_cull = 1 if no_footprint_cull else 0
    o.blend_impl = int(blend_impl)
    o.draw_limit = int(draw_limit)
    o.output_dtype = 1 if output_bf16 else 0  # frame stored as bfloat16; accumulation stays fp32
    o.depth_sort_passes = int(depth_sort_passes)  # 0: no bound (Rasterizer.render / render_batch fill in what the frames' counters have taught them)
    o.accum_dtype = 1 if accum_bf16 else 0        # configs[2] as worded: bf16 accumulators (measurement option, plain-C kernel)
    o.keep_flags = 1 if keep_flags else 0         # Rasterizer.enqueue sets it itself for the frames after the first since the last stats()
    o.saturation_rule = int(saturation_rule)      # 0: a quadrant stops once its colour cannot change (exact); 1: once T == 0.0f (the A/B reference)
    o.fine_binning = 1 if fine_binning else 0     # A/B switches (same frames): per-tile pairs; 1/2 = whole-frame / three-phase shard
    o.shard_preprocess = int(shard_preprocess)    # preprocess; tile bound of the pipelined blend walk (-1 = never); dense-wave SH threshold
    o.blend_pipe_tiles = int(blend_pipe_tiles)
    o.sh_dense_min = int(sh_dense_min)
    o.no_order_hint = 1 if no_order_hint else 0   # blend launch order by list length alone (default: by what each tile staged last frame)
    o.colour_stage = int(colour_stage)            # 0: sh_to_rgb when a tile first stages the gaussian (blend); 1: for every visible gaussian (preprocess)
    o.batch_views = int(batch_views)              # render_batch: at most this many views per launch sequence (0: as many as the workspace has slices)
    o.tile_row_block = int(tile_row_block)        # tile-row shards: 0 / 1 = single rows interleave, 2 = pairs of rows (whole 32x32 cell rows)
    return o


def shard_row_list(height: int, begin: int, step: int, block: int = 1):
    """The tile rows a tile-row shard owns, ascending = its strip rows (GsrOptions.tile_row_begin / _step / _block: blocks of `block`
    consecutive tile rows, block b is the shard's when b % step == begin)."""
    tiles_y = (height + TILE - 1) // TILE
    block = 2 if block == 2 else 1
    return [t for t in range(tiles_y) if (t // block) % max(step, 1) == (begin if step > 1 else 0)]


def shard_rows(height: int, begin: int, step: int, block: int = 1) -> int:
    return len(shard_row_list(height, begin, step, block))


class Rasterizer:
    """Owns the scratch workspace for one scene and renders frames of it.

    `views` > 1 makes the workspace that many slices (each a complete one-view workspace, gsr_workspace_bytes): render_batch /
    enqueue_batch then put `views` cameras at a time through ONE launch sequence of libgsr (gsr_render_batch: one preprocess that
    reads the scene once for all of them, one set of sorts, one blend — a quarter of the dispatches per frame at views = 4, each four
    times better filled).  Frames are bit-identical to single-view renders.  Single frames (render / enqueue) use slice 0."""

    def __init__(self, scene: GaussianScene, max_pairs: Optional[int] = None, views: int = 1):
        self.scene = scene
        self.views = max(1, min(int(views), _lib.GSR_MAX_BATCH_VIEWS))
        self.max_pairs = min(_lib.GSR_MAX_PAIRS, int(max_pairs) if max_pairs else max(1 << 20, 8 * scene.n))
        # radix passes the depth sort of this scene's frames has needed so far (GsrStats.sort_passes, learned whenever the counters
        # are read): passed as GsrOptions.depth_sort_passes so that the passes a frame does not need are not even enqueued
        self.sort_passes = 0
        self._ws: Optional[torch.Tensor] = None
        self._last_empty = False   # the last enqueue was a shard without tile rows: no kernel ran, its counters are all zero
        self._chained = False      # frames have been enqueued since the last stats(): the next one keeps their overflow record
        self._used = 1             # slices that hold unchecked frames (stats() reads them all)
        self._ws_key = None
        self._slice = 0            # bytes per slice of the workspace
        self.last_stats: Optional[Dict[str, int]] = None
        self.last_slice_stats: list = []   # stats() per slice: the last view each slice rendered

    # -- workspace ------------------------------------------------------------------------------
    def _workspace(self, width: int, height: int) -> torch.Tensor:
        key = (self.scene.n, width, height, self.max_pairs, self.views)
        if self._ws is None or self._ws_key != key:
            nbytes = _lib.workspace_bytes(self.scene.n, width, height, self.max_pairs)
            self._ws = None  # free the old one first
            self._ws = torch.empty(nbytes * self.views, dtype=torch.uint8, device=self.scene.device)
            assert self._ws.data_ptr() % 256 == 0 and nbytes % 256 == 0
            # libgsr needs no initialisation (every frame clears its control block); the head of every slice is zeroed so that a
            # gsr_read_stats BEFORE any frame has run there reads zeros rather than whatever the allocator left, and so that a
            # batch chained behind unchecked single frames (keep_flags) finds an empty record in the slices they never used
            for v in range(self.views):
                self._ws[v * nbytes: v * nbytes + min(4096, nbytes)].zero_()
            self._ws_key, self._slice = key, nbytes
            self._chained = self._last_empty = False
            self._used = 1
        return self._ws

    def _out_shape(self, cam: GsrCamera, opts: GsrOptions):
        if opts.output_layout == 0:
            return (cam.height, cam.width, 3), (cam.height, cam.width)
        if opts.output_layout == 1:
            return (cam.width, cam.height, 3), (cam.width, cam.height)
        rows = shard_rows(cam.height, opts.tile_row_begin, opts.tile_row_step, opts.tile_row_block) * TILE
        return (rows, cam.width, 3), (rows, cam.width)

    def bounded(self, opts: Optional[GsrOptions] = None) -> GsrOptions:
        """opts with the learned depth-sort bound filled in (a copy), unless the caller set one.  render() / render_batch()
        apply it themselves (they check every frame and re-render); for enqueue() / FramesInFlight.submit() the caller opts
        in with this — the frames are then unchecked until the next stats(), which reports ANY of them that exceeded it."""
        opts = opts or make_options()
        if opts.depth_sort_passes != 0 or self.sort_passes == 0:
            return opts
        o = GsrOptions.from_buffer_copy(opts)
        o.depth_sort_passes = self.sort_passes
        return o

    # -- one frame ------------------------------------------------------------------------------
    def enqueue(self, cam: GsrCamera, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None,
                final_T: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Enqueue one frame on the current stream with `opts` as given; no host synchronisation and no check of the
        caller's bounds (max_pairs; opts.depth_sort_passes if set).  The frames enqueued since the last stats() are chained
        with GsrOptions.keep_flags, so the next stats() / render() reports a bound exceeded by ANY of them, not only the last."""
        opts = opts or make_options()
        ws = self._workspace(cam.width, cam.height)
        shape, _ = self._out_shape(cam, opts)
        dtype = torch.bfloat16 if opts.output_dtype == 1 else torch.float32
        if out is None:
            # strips may include rows below the frame's last pixel row: keep them defined
            out = torch.zeros(shape, dtype=dtype, device=self.scene.device) if opts.output_layout == 2 else \
                torch.empty(shape, dtype=dtype, device=self.scene.device)
        elif tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous() or not out.is_cuda:
            raise ValueError(f"out must be a contiguous {dtype} CUDA tensor of shape {shape}")
        if out.numel() == 0:  # a shard that owns no tile row (more ranks than tile rows): nothing to render
            self._last_empty = True
            return out
        if self._chained and not opts.keep_flags:
            opts = GsrOptions.from_buffer_copy(opts)
            opts.keep_flags = 1
        sc = self.scene.c_struct()
        tptr = final_T.data_ptr() if final_T is not None else None
        check(lib.gsr_render_forward(C.byref(sc), C.byref(cam), C.byref(opts), self.max_pairs, ws.data_ptr(), ws.numel(),
                                     out.data_ptr(), tptr, _stream_ptr(self.scene.device)))
        self._chained, self._last_empty = True, False
        return out

    def stats(self) -> Dict[str, int]:
        """Counters of the last enqueued frame (synchronises the stream); its overflow record covers every frame since the
        previous stats().  Raises GsrPairOverflow / GsrSortPasses when one of them exceeded a bound."""
        if self._ws is None or self._last_empty:  # no workspace yet / an empty shard: no kernel ran, nothing to read
            self.last_stats = {k: 0 for k, _ in GsrStats._fields_ if not k.startswith("_")}
            self.last_slice_stats = [self.last_stats]
            return self.last_stats
        # every slice that holds unchecked frames (one, unless batches ran): the last view each rendered, its record sticky over all of them
        per, worst_rc = [], 0
        for v in range(self._used):
            st = GsrStats()
            rc = lib.gsr_read_stats(self._ws.data_ptr() + v * self._slice, self._slice, C.byref(st), _stream_ptr(self.scene.device))
            per.append(st.as_dict())
            self.sort_passes = max(self.sort_passes, int(st.sort_passes))  # also when a frame was short of passes: the retry has them
            if rc == _lib.GSR_ERR_PAIR_OVERFLOW or (rc != 0 and worst_rc != _lib.GSR_ERR_PAIR_OVERFLOW):
                worst_rc = rc
        self._chained, self._used = False, 1  # the next frame starts from a cleared control block
        self.last_slice_stats = per
        self.last_stats = dict(per[0])
        if worst_rc:  # what a re-render needs: the worst slice's figures
            self.last_stats["n_pairs_bbox"] = max(d["n_pairs_bbox"] for d in per)
            self.last_stats["sort_passes"] = max(d["sort_passes"] for d in per)
            self.last_stats["overflow"] = 0
            for d in per:
                self.last_stats["overflow"] |= d["overflow"]
        check(worst_rc)
        return self.last_stats

    @staticmethod
    def _incomplete(what: str, stats) -> "_lib.GsrError":
        """The error of a frame that is still over a bound after every retry: the status the LAST attempt reported
        (pair overflow before sort passes), never GSR_ERR_BAD_ARG — no argument was bad."""
        ov = int((stats or {}).get("overflow", 0))
        if ov & 1 or not ov & 2:
            return _lib.GsrPairOverflow(_lib.GSR_ERR_PAIR_OVERFLOW, f"{what}: {stats}")
        return _lib.GsrSortPasses(_lib.GSR_ERR_SORT_PASSES, f"{what}: {stats}")

    def _grow_pairs(self, slack_div: int) -> None:
        """After GsrPairOverflow: room for what the worst frame needed, or give up when that cannot be had."""
        need = int(self.last_stats["n_pairs_bbox"])
        if need >= _lib.GSR_MAX_PAIRS:
            raise _lib.GsrError(_lib.GSR_ERR_PAIR_OVERFLOW, f"the frame needs {need} pairs, more than libgsr can index")
        grown = int(min(_lib.GSR_MAX_PAIRS, need + need // slack_div + 1024))
        if grown <= self.max_pairs:
            raise _lib.GsrError(_lib.GSR_ERR_PAIR_OVERFLOW, f"pair overflow persists at max_pairs = {self.max_pairs} (need {need})")
        self.max_pairs = grown

    def render(self, cam: GsrCamera, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None,
               return_T: bool = False):
        """Render one frame and verify it is complete: grows the pair buffer / raises the learned depth-sort bound and
        re-renders when the frame exceeded one (at most MAX_RETRIES times)."""
        opts = opts or make_options()
        unbounded, tried = False, -1
        for _ in range(MAX_RETRIES + 1):
            final_T = None
            if return_T:
                _, tshape = self._out_shape(cam, opts)
                final_T = torch.ones(tshape, dtype=torch.float32, device=self.scene.device)
            # (frames enqueued before this one and not yet checked share its overflow record: if one of THEM exceeded a bound,
            # this frame is re-rendered once with room for it — nothing is hidden and nothing is wrong)
            img = self.enqueue(cam, opts if unbounded else self.bounded(opts), out, final_T)
            try:
                self.stats()
            except _lib.GsrPairOverflow:
                self._grow_pairs(8)
                continue
            except _lib.GsrSortPasses:
                if opts.depth_sort_passes != 0:
                    raise  # the caller's own bound
                unbounded = self.sort_passes <= tried  # stats() raises the learned bound; if it did not grow, enqueue every pass
                tried = self.sort_passes
                continue
            return (img, final_T) if return_T else img
        raise self._incomplete(f"frame still incomplete after {MAX_RETRIES} re-renders", self.last_stats)

    def _batch_out(self, cams, opts: GsrOptions, out: Optional[torch.Tensor]):
        """(out, frame_stride in elements) of a batch: whole frames [B,H,W,3], or with a tile-row shard (output_layout = 2) the
        strips [B,rows*16,W,3]."""
        if opts.output_layout == 1:
            raise ValueError("batches render [H,W,3] frames or shard strips")
        shape, _ = self._out_shape(cams[0], opts)
        full = (len(cams),) + tuple(shape)
        dtype = torch.bfloat16 if opts.output_dtype == 1 else torch.float32
        frame = shape[0] * shape[1] * shape[2]
        if out is None:
            out = (torch.zeros if opts.output_layout == 2 else torch.empty)(full, dtype=dtype, device=self.scene.device)
        elif (tuple(out.shape) != full or out.dtype != dtype or not out.is_cuda or (frame and not out[0].is_contiguous())
              or (len(cams) > 1 and out.stride(0) < frame)):
            # (consecutive frames may lie further apart than a frame: the strips of a padded wire buffer, dist.FrameGather)
            raise ValueError(f"out must be a {dtype} CUDA tensor of shape {full} whose frames are contiguous")
        return out, (out.stride(0) if len(cams) > 1 and frame else frame)

    def enqueue_batch(self, cams, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Enqueue several views on the current stream, `views` at a time through one launch sequence (gsr_render_batch), with `opts`
        as given: no host synchronisation, no check of the caller's bounds — like enqueue(), the next stats() speaks for all of them."""
        opts = opts or make_options()
        cams = list(cams)
        if not cams:
            raise ValueError("a batch needs at least one view")
        out, stride = self._batch_out(cams, opts, out)
        if out.numel() == 0:  # a shard that owns no tile row
            self._last_empty = True
            return out
        arr = (GsrCamera * len(cams))(*cams)
        ws = self._workspace(cams[0].width, cams[0].height)
        o = GsrOptions.from_buffer_copy(opts)
        # a slice's first view clears its record (libgsr chains its later views) — unless frames enqueued before are still unchecked
        o.keep_flags = 1 if (self._chained or opts.keep_flags) else 0
        sc = self.scene.c_struct()
        check(lib.gsr_render_batch(C.byref(sc), arr, len(cams), C.byref(o), self.max_pairs, ws.data_ptr(), ws.numel(),
                                   out.data_ptr(), stride, _stream_ptr(self.scene.device)))
        per_launch = self.views if o.batch_views == 0 else min(self.views, o.batch_views)
        self._used = max(self._used, min(per_launch, len(cams)))
        self._chained, self._last_empty = True, False
        return out

    def render_batch(self, cams, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Several views of the resident scene in one call: [B,H,W,3] (a tile-row shard: [B,rows*16,W,3]), `views` of them per launch
        sequence.  Pair buffers and the depth-sort bound are sized on the fly: a view that exceeds one makes the batch re-run with
        room for the worst view (at most MAX_RETRIES times)."""
        opts = opts or make_options()
        cams = list(cams)
        unbounded = False
        for _ in range(MAX_RETRIES + 1):
            o = opts if unbounded else self.bounded(opts)
            out = self.enqueue_batch(cams, o, out)
            try:
                self.stats()
                return out
            except _lib.GsrPairOverflow:
                self._grow_pairs(4)
            except _lib.GsrSortPasses:
                if opts.depth_sort_passes != 0:
                    raise
                unbounded = o.depth_sort_passes >= self.sort_passes  # the reported need did not exceed what was enqueued: play safe
        raise self._incomplete(f"batch still incomplete after {MAX_RETRIES} re-renders", self.last_stats)

    def fit_pairs(self, cam: GsrCamera, opts: Optional[GsrOptions] = None, slack: float = 1.25) -> int:
        """Size the pair buffers to this view: one probing frame, then max_pairs = slack * D (+ margin).
        Sort grids and histogram tables scale with max_pairs, so a snug bound is also the fast one."""
        self.render(cam, opts)
        need = int(self.last_stats["n_pairs_bbox"])
        self.max_pairs = int(min(_lib.GSR_MAX_PAIRS, max(4096, slack * need + 4096)))
        return self.max_pairs

    # -- stage-by-stage (tests, helper functions) -------------------------------------------------
    def preprocess_debug(self, cam: GsrCamera, opts: Optional[GsrOptions] = None) -> Dict[str, torch.Tensor]:
        """Run stage 1 alone and return every per-gaussian intermediate the reference's helpers produce, indexed like the
        file the scene was loaded from (whatever order the scene is stored in)."""
        opts = opts or make_options()
        n, dev = self.scene.n, self.scene.device
        f32, i64 = torch.float32, torch.int64
        out = {
            "cov3d": torch.empty((n, 3, 3), dtype=f32, device=dev), "cam_means": torch.empty((n, 3), dtype=f32, device=dev),
            "cov2d": torch.empty((n, 2, 2), dtype=f32, device=dev), "screen_means": torch.empty((n, 2), dtype=f32, device=dev),
            "tile_bboxes": torch.empty((n, 4), dtype=i64, device=dev), "sigmas": torch.empty((n, 3), dtype=f32, device=dev),
            "pixel_bboxes": torch.empty((n, 4), dtype=i64, device=dev), "rgb": torch.empty((n, 3), dtype=f32, device=dev),
            "opacity": torch.empty((n,), dtype=f32, device=dev),
        }
        dbg = GsrDebugOut()
        for k, v in out.items():
            setattr(dbg, k, v.data_ptr())
        ws = self._workspace(cam.width, cam.height)
        sc = self.scene.c_struct()
        check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), C.byref(dbg),
                                 _stream_ptr(dev)))
        if self.scene.order_t is not None:  # the library works in scene order; callers (and the reference's helpers) index by file order
            for k, v in out.items():
                back = torch.empty_like(v)
                back[self.scene.order_t] = v
                out[k] = back
        return out


class FramesInFlight:
    """Throughput mode for independent frames (a camera path, a batch of views): `slots` frames in flight, each on its own
    HIP stream with its own workspace, so that one frame's HBM-bound stages (preprocess, sorts) overlap another's
    VALU-bound blend and the short launch-bound kernels of a multi-GPU shard overlap each other.  libgsr keeps no state and
    every call is stream-asynchronous, so this is plain use of the C ABI: one (workspace, stream) pair per slot.  Frames
    are bit-identical to single-stream rendering; only the time per frame changes (tools/stream_overlap.py).

    submit() returns the slot it used; the caller owns the ordering of its output buffers: wait(slot) makes the current
    stream wait for that slot's last frame."""

    def __init__(self, scene: GaussianScene, slots: int = 4, max_pairs: Optional[int] = None, views: int = 1):
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.scene = scene
        self.rasterizers = [Rasterizer(scene, max_pairs=max_pairs, views=views) for _ in range(slots)]
        self.streams = [torch.cuda.Stream(device=scene.device) for _ in range(slots)]
        cur = torch.cuda.current_stream(scene.device)
        for st in self.streams:  # the scene upload (and whatever else the caller enqueued) comes first
            st.wait_stream(cur)
        self._next = 0

    @property
    def slots(self) -> int:
        return len(self.rasterizers)

    def set_max_pairs(self, max_pairs: int) -> None:
        for r in self.rasterizers:
            r.max_pairs = int(max_pairs)

    def set_sort_passes(self, passes: int) -> None:
        """Share the depth-sort bound one slot has learned (Rasterizer.sort_passes) with all of them."""
        for r in self.rasterizers:
            r.sort_passes = max(r.sort_passes, int(passes))

    def submit(self, cam: GsrCamera, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None,
               slot: Optional[int] = None) -> int:
        """Enqueue one frame on the next slot's stream (round robin) and return the slot.  Unchecked, with `opts` as given
        (Rasterizer.enqueue): stats(slot) afterwards speaks for every frame the slot has rendered since its last stats()."""
        k = self._next if slot is None else int(slot)
        if slot is None:
            self._next = (self._next + 1) % len(self.rasterizers)
        with torch.cuda.stream(self.streams[k]):
            self.rasterizers[k].enqueue(cam, opts, out=out)
        return k

    def submit_batch(self, cams, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None, slot: Optional[int] = None) -> int:
        """submit() for several views at once: Rasterizer.enqueue_batch on the next slot's stream (`views` of them per launch sequence)."""
        k = self._next if slot is None else int(slot)
        if slot is None:
            self._next = (self._next + 1) % len(self.rasterizers)
        with torch.cuda.stream(self.streams[k]):
            self.rasterizers[k].enqueue_batch(cams, opts, out=out)
        return k

    def render_batch(self, cams, opts: Optional[GsrOptions] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rasterizer.render_batch with the views spread round robin over the slots (gsr_render_batch_slots; with `views` > 1 each
        slot takes `views` consecutive cameras per launch sequence): [B,H,W,3],
        bit-identical to the single-stream batch.  The current stream waits for every slot before this returns; a view that
        overflows the pair buffers makes the batch re-run with room for it."""
        opts = opts or make_options()
        if opts.output_layout != 0 or opts.tile_row_step > 1:
            raise ValueError("render_batch renders whole [H,W,3] frames")
        cams = list(cams)
        if not cams:
            raise ValueError("render_batch needs at least one view")
        arr = (GsrCamera * len(cams))(*cams)
        W, H = cams[0].width, cams[0].height
        dev = self.scene.device
        dtype = torch.bfloat16 if opts.output_dtype == 1 else torch.float32
        if out is None:
            out = torch.empty((len(cams), H, W, 3), dtype=dtype, device=dev)
        elif tuple(out.shape) != (len(cams), H, W, 3) or out.dtype != dtype or not out.is_contiguous() or not out.is_cuda:
            raise ValueError(f"out must be a contiguous {dtype} CUDA tensor of shape {(len(cams), H, W, 3)}")
        sc = self.scene.c_struct()
        n = len(self.rasterizers)
        cur = torch.cuda.current_stream(dev)
        r0 = self.rasterizers[0]
        unbounded = False
        for _ in range(MAX_RETRIES + 1):
            wss = [r._workspace(W, H) for r in self.rasterizers]  # all of one size: the slots share max_pairs
            for st in self.streams:
                st.wait_stream(cur)  # `out` (and the workspaces) may have been allocated / used on the current stream
            ws_arr = (C.c_void_p * n)(*[w.data_ptr() for w in wss])
            st_arr = (C.c_void_p * n)(*[int(st.cuda_stream) for st in self.streams])
            self.set_sort_passes(max(r.sort_passes for r in self.rasterizers))
            o = GsrOptions.from_buffer_copy(opts if unbounded else r0.bounded(opts))
            for k in range(n):  # frames submit()ted and not yet checked: read (and report) their record before the batch clears it
                if self.rasterizers[k]._chained:
                    self.stats(k)
            o.keep_flags = 0  # a slot's first view clears its record, libgsr chains the slot's later views
            check(lib.gsr_render_batch_slots(C.byref(sc), arr, len(cams), C.byref(o), r0.max_pairs, ws_arr, wss[0].numel(), st_arr, n,
                                             out.data_ptr(), H * W * 3))
            per_launch = min(r0.views if o.batch_views == 0 else min(r0.views, o.batch_views), len(cams))
            groups = (len(cams) + per_launch - 1) // per_launch
            used = min(n, groups)
            sizes = [min(per_launch, len(cams) - g * per_launch) for g in range(groups)]
            for k, r in enumerate(self.rasterizers[:used]):
                r._chained, r._last_empty, r._used = True, False, max(sizes[k::n])  # slices its groups rendered into
            need, short = 0, False
            for k in range(used):
                try:
                    self.stats(k)
                except _lib.GsrPairOverflow:
                    need = max(need, int(self.rasterizers[k].last_stats["n_pairs_bbox"]))
                except _lib.GsrSortPasses:
                    if opts.depth_sort_passes != 0:
                        raise
                    short = True  # stats() has raised that slot's learned bound; the re-run shares it
            for k in range(n):
                self.wait(k)
            if need == 0 and not short:
                return out
            if need:
                if need >= _lib.GSR_MAX_PAIRS:
                    raise _lib.GsrError(_lib.GSR_ERR_PAIR_OVERFLOW, f"a view needs {need} pairs, more than libgsr can index")
                self.set_max_pairs(int(min(_lib.GSR_MAX_PAIRS, need + need // 4 + 1024)))
            if short:
                unbounded = o.depth_sort_passes >= max(r.sort_passes for r in self.rasterizers)  # the need did not grow: play safe
        raise Rasterizer._incomplete(f"batch still incomplete after {MAX_RETRIES} re-renders",
                                     max((r.last_stats or {} for r in self.rasterizers), key=lambda d: d.get("overflow", 0)))

    def wait(self, slot: int) -> None:
        torch.cuda.current_stream(self.scene.device).wait_stream(self.streams[slot])

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()

    def stats(self, slot: int = 0) -> Dict[str, int]:
        with torch.cuda.stream(self.streams[slot]):
            return self.rasterizers[slot].stats()
