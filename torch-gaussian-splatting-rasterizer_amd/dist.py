"""Tile-row sharding of one frame over the GPUs of a node + the framebuffer gather.

The reference has no distributed code.  A frame shards naturally by 16x16 tile rows: every pixel depends only
on the gaussians binned to its tile, so a rank bins and blends every G-th PAIR of tile rows — 2b, 2b+1 for b = G-1-r, 2G-1-r, ...: whole rows of
the 32x32 cells the binning works in, interleaved for load balance (TileRowPlan; single rows r, r+G, ... until round 4) — into a
compact strip [rows_r*16, W, 3], and ONE collective — a gather to rank 0 (RCCL over xGMI on the
GPU box: each peer->root transfer rides its own link) — exchanges the strips.  Per-pixel blend order does not
depend on the sharding, so the assembled frame is bit-identical to the single-GPU frame.

Everything here works on any `torch.distributed` backend and device (gloo/CPU in the tests, nccl(=RCCL)/GPU in
bench.py); the strips themselves come from `Rasterizer.enqueue(..., output_layout=2)`.
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

TILE = 16


class TileRowPlan:
    """Which tile rows each rank owns, and where its strip rows land in the frame.

    `block` = 2 (default): PAIRS of rows — the two tile rows of one 32x32 binning cell — interleaved (rank r: blocks b = world - 1 - r,
    2 world - 1 - r, ..., i.e. rows 2b, 2b + 1; GsrOptions.tile_row_block): a rank bins and sorts only the cells it owns, and fewer gaussians reach it.  `block` = 1:
    single tile rows (rank r: rows world - 1 - r, 2 world - 1 - r, ...; rounds 1-4 with r in place of world - 1 - r): two ranks share every cell row, each emitting and sorting all its
    pairs; the ranks' row counts then differ by one at most instead of two.  Same frame either way.  Measured, bicycle stand-in at 1080p,
    slowest rank, ms per frame (tools/shard_timing.py, block 1 / 2): G = 2: 0.368 / 0.326, 4: 0.247 / 0.224, 8: 0.170 / 0.152."""

    def __init__(self, height: int, width: int, world: int, block: int = 2):
        if world < 1:
            raise ValueError("world must be >= 1")
        if block not in (1, 2):
            raise ValueError("block must be 1 (single tile rows) or 2 (pairs of rows)")
        self.height, self.width, self.world, self.block = int(height), int(width), int(world), int(block)
        self.tiles_y = (self.height + TILE - 1) // TILE
        # rank r takes the blocks b with b % world == world - 1 - r: the blocks left over when world does not divide their number go to
        # the LAST ranks, so that rank 0 — which also receives and de-interleaves everybody's strips — never carries an extra one
        self.begin = [self.world - 1 - r for r in range(self.world)]
        self.rows = [[t for t in range(self.tiles_y) if (t // self.block) % self.world == self.begin[r]] for r in range(self.world)]
        self.max_rows = max(1, max(len(x) for x in self.rows))
        # frame tile row t <- strip row index_of[t] of rank owner[t]
        self.owner = [self.world - 1 - (t // self.block) % self.world for t in range(self.tiles_y)]
        self.index_of = [self.rows[self.owner[t]].index(t) for t in range(self.tiles_y)]

    def strip_shape(self, rank: int):
        """Shape of rank's compact strip as libgsr writes it (output_layout = 2)."""
        return (len(self.rows[rank]) * TILE, self.width, 3)

    def padded_shape(self):
        """Common shape used on the wire (equal-size gather)."""
        return (self.max_rows * TILE, self.width, 3)

    def shard_options(self, rank: int):
        return dict(tile_row_begin=self.begin[rank], tile_row_step=self.world, output_layout=2, tile_row_block=self.block)

    def _scatter_rows(self, grid: torch.Tensor, strips: List[torch.Tensor]):
        for r, s in enumerate(strips):
            k = len(self.rows[r])
            if k:
                rows = s[: k * TILE].view(k, TILE, self.width, 3)
                if self.block == 1:
                    grid[self.begin[r]::self.world] = rows
                else:
                    grid[torch.as_tensor(self.rows[r], device=grid.device)] = rows

    def assemble(self, strips: List[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """strips[r]: padded strip of rank r -> frame [H,W,3]."""
        ref = strips[0]
        padded_h = self.tiles_y * TILE
        if out is None:
            out = torch.empty((self.height, self.width, 3), dtype=ref.dtype, device=ref.device)
        if padded_h == self.height:
            self._scatter_rows(out.view(self.tiles_y, TILE, self.width, 3), strips)
            return out
        full = torch.empty((padded_h, self.width, 3), dtype=ref.dtype, device=ref.device)
        self._scatter_rows(full.view(self.tiles_y, TILE, self.width, 3), strips)
        out.copy_(full[: self.height])
        return out

    def split(self, frame: torch.Tensor, rank: int) -> torch.Tensor:
        """Inverse of assemble for one rank (tests): the padded strip rank would produce from `frame`."""
        padded_h = self.tiles_y * TILE
        full = frame.new_zeros((padded_h, self.width, 3))
        full[: self.height] = frame
        strip = frame.new_zeros(self.padded_shape())
        k = len(self.rows[rank])
        if k:
            idx = torch.as_tensor(self.rows[rank], device=frame.device)
            strip[: k * TILE] = full.view(self.tiles_y, TILE, self.width, 3)[idx].reshape(k * TILE, self.width, 3)
        return strip


class FrameGather:
    """Pre-allocated equal-size gather of the strips to rank 0, double-buffered so that the gather of frame k
    (RCCL, on the communicator's own stream) can overlap the render of frame k+1.

    `views` > 1: every wire buffer holds the strips of `views` consecutive frames ([views, rows, W, 3]) — what a rank renders through
    ONE launch sequence (Rasterizer.enqueue_batch) — and ONE collective moves them all (views x the message, the same number of
    messages per frame batch); rank 0's `frame` is then [views, H, W, 3]."""

    def __init__(self, plan: TileRowPlan, rank: int, device, dtype=torch.float32, group=None, buffers: int = 2, views: int = 1):
        self.plan, self.rank, self.group, self.views = plan, rank, group, int(views)
        V = self.views
        self._strips = [torch.zeros((V,) + plan.padded_shape(), dtype=dtype, device=device) for _ in range(buffers)]
        self.recvs = self._recv_all = self._row_src = None
        self.frame = None
        if rank == 0:
            # the frames live in a buffer padded to whole tile rows; `frame` is its first H pixel rows
            self._frame_padded = torch.zeros((V, plan.tiles_y * TILE, plan.width, 3), dtype=dtype, device=device)
            self.frame = self._frame_padded[0, : plan.height] if V == 1 else self._frame_padded[:, : plan.height]
            if plan.world > 1:
                # one receive buffer per wire buffer, the gather list are views into it, so that de-interleaving the
                # G x V strips is ONE index_select (one kernel, one Python call per batch on the root) instead of G x V copies
                self._recv_all = [torch.zeros((plan.world, V) + plan.padded_shape(), dtype=dtype, device=device) for _ in range(buffers)]
                self.recvs = [[ra[r] for r in range(plan.world)] for ra in self._recv_all]
                k = torch.arange(V)
                # (view k, frame tile row t) <- wire row of rank owner[t], view k, its strip row index_of[t]
                src = (torch.as_tensor(plan.owner)[None, :] * V + k[:, None]) * plan.max_rows + torch.as_tensor(plan.index_of)[None, :]
                self._row_src = src.reshape(-1).to(device)

    @property
    def strips(self):
        return [s[0] for s in self._strips] if self.views == 1 else self._strips

    @property
    def strip(self) -> torch.Tensor:
        return self.strips[0]

    def own_view(self, buf: int = 0) -> torch.Tensor:
        """The leading rows of wire buffer `buf` that libgsr writes this rank's strip(s) into: [rows, W, 3], or with `views` > 1
        [views, rows, W, 3] (a strided view: consecutive strips lie a padded strip apart)."""
        k = len(self.plan.rows[self.rank])
        return self._strips[buf][0, : k * TILE] if self.views == 1 else self._strips[buf][:, : k * TILE]

    def _host_staged(self) -> bool:
        """gloo cannot move device memory: GPU strips go through host buffers (rehearsal of the multi-rank path on a
        box without RCCL peers; the production backend is nccl = RCCL, device to device)."""
        return self._strips[0].is_cuda and dist.get_backend(self.group) == "gloo"

    def gather_async(self, buf: int = 0):
        """Start the collective for wire buffer `buf` (every rank calls it once its strip is enqueued on the
        current stream).  Returns a handle for finish()."""
        if self.plan.world == 1:
            return (buf, None, None)
        if self._host_staged():
            send = self._strips[buf].cpu()  # synchronises with the render of this strip
            recv = [torch.empty_like(send) for _ in range(self.plan.world)] if self.rank == 0 else None
            work = dist.gather(send, recv, dst=0, group=self.group, async_op=True)
            return (buf, work, recv)
        work = dist.gather(self._strips[buf], self.recvs[buf] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        return (buf, work, None)

    def finish(self, handle) -> Optional[torch.Tensor]:
        """Wait for a gather (stream-ordered on GPU backends) and, on rank 0, de-interleave into the frame(s)."""
        buf, work, host = handle
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        if self.plan.world == 1:
            for k in range(self.views):
                self.plan.assemble([self._strips[buf][k]], self._frame_padded[k, : self.plan.height])
            return self.frame
        if host is not None:
            for dst, src in zip(self.recvs[buf], host):
                dst.copy_(src)
        row = TILE * self.plan.width * 3
        torch.index_select(self._recv_all[buf].view(self.plan.world * self.views * self.plan.max_rows, row), 0, self._row_src,
                           out=self._frame_padded.view(self.views * self.plan.tiles_y, row))
        return self.frame

    def gather(self, buf: int = 0) -> Optional[torch.Tensor]:
        """Blocking form: every rank calls it after its strip is complete.  Rank 0 gets the frame."""
        return self.finish(self.gather_async(buf))


class ShardedFrames:
    """The N-GPU frame loop of one rank with `slots` frames (or, with `views` > 1, batches of `views` frames) in flight.

    Frame f renders on stream f % slots into wire buffer f % slots (`render(slot, cam, strip)` enqueues this rank's strip
    on the CURRENT stream — renderer.FramesInFlight's rasterizer of that slot); its strip is gathered asynchronously (RCCL
    stream, ordered behind the render stream) while the next frames render.  Frames are finished in order on the caller's
    stream, at most slots - 1 behind.  Before buffer f % slots is reused its render stream waits for the caller's stream,
    which by then has waited for the previous gather out of that buffer and, on rank 0, has de-interleaved the receive
    buffer that gather f will overwrite.  With streams = None (CPU tensors, or one stream) everything runs in program
    order on the current stream; the schedule is the same.
    With `views` > 1 a submission is a LIST of up to `views` cameras, `render(slot, cams, strips)` gets the [views, rows, W, 3] view of
    the wire buffer (Rasterizer.enqueue_batch(cams, opts, out=strips[:len(cams)]) renders them through one launch sequence), and one
    gather moves the batch.

    submit() returns the frame(s) finished by this call on rank 0 (a view of FrameGather.frame, overwritten by the next
    finish) or None; drain() finishes what is still in flight and returns the last frame(s)."""

    def __init__(self, plan: TileRowPlan, rank: int, device, slots: int, render: Callable, dtype=torch.float32, group=None,
                 streams=None, views: int = 1):
        if slots < 1:
            raise ValueError("slots must be >= 1")
        if streams is not None and len(streams) != slots:
            raise ValueError("one stream per slot")
        self.plan, self.rank, self.device, self.slots = plan, rank, device, int(slots)
        self.fg = FrameGather(plan, rank, device, dtype=dtype, group=group, buffers=max(2, self.slots), views=views)
        self.render, self.streams = render, streams
        self.pending: List = []
        self.frames_submitted = 0

    def _finish_oldest(self):
        _, h = self.pending.pop(0)
        return self.fg.finish(h)

    def submit(self, cam):
        f = self.frames_submitted
        self.frames_submitted += 1
        k = f % self.slots
        done = None
        while self.pending and self.pending[0][0] <= f - self.slots:
            done = self._finish_oldest()
        if self.streams is not None:
            st = self.streams[k]
            st.wait_stream(torch.cuda.current_stream(self.device))
            ctx = torch.cuda.stream(st)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            self.render(k, cam, self.fg.own_view(k))
            h = self.fg.gather_async(k)
        self.pending.append((f, h))
        return done

    def drain(self):
        out = None
        while self.pending:
            out = self._finish_oldest()
        return out
