"""CPU, world_size 2 (and 3) over gloo: the tile-row shard plan, the framebuffer gather and the reassembly
(gsr_amd/dist.py) — the N>1 data path of bench.py minus the HIP render that fills the strips."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    import sys

    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsr_amd import dist as gdist

    g = torch.Generator().manual_seed(1234)
    frame = torch.rand((H, W, 3), generator=g)                   # same "rendered frame" on every rank
    plan = gdist.TileRowPlan(H, W, world, 1 + (H // 16) % 2)    # single rows or pairs of rows, by the frame
    fg = gdist.FrameGather(plan, rank, "cpu")
    own = fg.own_view()
    assert tuple(own.shape) == plan.strip_shape(rank)
    own.copy_(plan.split(frame, rank)[: own.shape[0]])           # what libgsr would write (output_layout = 2)
    out = fg.gather()
    ok = bool(torch.equal(out, frame)) if rank == 0 else out is None
    # double-buffered asynchronous form (what bench.py runs): two frames in flight
    frame2 = frame.flip(0).contiguous()
    own1 = fg.own_view(1)
    own1.copy_(plan.split(frame2, rank)[: own1.shape[0]])
    h0, h1 = fg.gather_async(0), fg.gather_async(1)
    a = fg.finish(h0)
    a = a.clone() if a is not None else None
    b = fg.finish(h1)
    ok = ok and (bool(torch.equal(a, frame) and torch.equal(b, frame2)) if rank == 0 else (a is None and b is None))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def _pipeline_worker(rank, world, port, H, W, q):
    """dist.ShardedFrames (bench.py's N-GPU loop) with 1, 3 and 6 (bench.py's default) frames in flight over gloo: a fake render writes
    the rank's strip of a known frame; 14 frames, so every wire buffer is reused; rank 0 must get every frame back, in order."""
    import sys

    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsr_amd import dist as gdist

    g = torch.Generator().manual_seed(99)
    truth = [torch.rand((H, W, 3), generator=g) for _ in range(14)]
    plan = gdist.TileRowPlan(H, W, world, 1 + (H // 16) % 2)    # single rows or pairs of rows, by the frame

    def render(slot, frame_index, strip):
        strip.copy_(plan.split(truth[frame_index], rank)[: strip.shape[0]])

    ok = True
    for slots in (1, 3, 6):
        sf = gdist.ShardedFrames(plan, rank, "cpu", slots, render)
        got = []
        for i in range(len(truth)):
            f = sf.submit(i)
            if f is not None:
                got.append(f.clone())
        while sf.pending:
            f = sf._finish_oldest()
            if f is not None:
                got.append(f.clone())
        if rank == 0:
            ok = ok and len(got) == len(truth) and all(torch.equal(a, b) for a, b in zip(got, truth))
        else:
            ok = ok and not got
    # batches of `views` frames per submission (a rank renders them through one launch sequence, one gather moves them): 14 frames in
    # batches of 3 — the last one holds 2 —, with 1 and 2 batches in flight
    def render_batch(slot, idxs, strips):
        assert strips.shape[0] == 3 and tuple(strips.shape[1:]) == plan.strip_shape(rank)
        for j, i in enumerate(idxs):
            strips[j].copy_(plan.split(truth[i], rank)[: strips.shape[1]])

    for slots in (1, 2):
        sf = gdist.ShardedFrames(plan, rank, "cpu", slots, render_batch, views=3)
        got, sizes = [], []
        batches = [list(range(i, min(i + 3, len(truth)))) for i in range(0, len(truth), 3)]
        for bt in batches:
            sizes.append(len(bt))
            f = sf.submit(bt)
            if f is not None:
                got.extend(f[: sizes[len(got) // 3]].clone().unbind(0))
        while sf.pending:
            f = sf._finish_oldest()
            if f is not None:
                got.extend(f[: sizes[len(got) // 3]].clone().unbind(0))
        if rank == 0:
            ok = ok and len(got) == len(truth) and all(torch.equal(a, b) for a, b in zip(got, truth))
        else:
            ok = ok and not got
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world,H,W", [(2, 93, 150), (3, 200, 64)])
def test_frames_in_flight_pipeline_returns_every_frame_in_order(world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,H,W", [(2, 96, 160), (2, 93, 150), (3, 200, 64)])
def test_gather_reassembles_the_frame(world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def test_plan_covers_every_tile_row_once():
    from gsr_amd import dist as gdist

    for H in (16, 17, 93, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            for block in (1, 2):
                plan = gdist.TileRowPlan(H, 64, world, block)
                rows = sorted(r for rs in plan.rows for r in rs)
                assert rows == list(range((H + 15) // 16))
                assert max(len(r) for r in plan.rows) - min(len(r) for r in plan.rows) <= block
                frame = torch.arange(H * 64 * 3, dtype=torch.float32).view(H, 64, 3)
                assert torch.equal(plan.assemble([plan.split(frame, r) for r in range(world)]), frame)
