#!/usr/bin/env python3
"""Multi-rank check of the tile-row sharding + framebuffer gather, launched under torch.distributed.run by
tests/test_gpu_configs.py:  every rank renders its interleaved tile rows of the same scene for several frames in a row
through the same schedule as bench.py (gsr_amd.dist.ShardedFrames: several frames in flight, asynchronous gathers), rank 0 assembles each frame and
compares it bit for bit with its own unsharded render.  Backend from GSR_BENCH_BACKEND: nccl (= RCCL, one rank per GPU) or
gloo (ranks share the GPUs that exist; strips are staged through the host).  Prints DIST_CHECK_OK on success."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local % ndev if backend == "gloo" else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)

    from gsr_amd import dist as gdist
    from gsr_amd import renderer, synthetic

    W, H, n = 1000, 600, int(os.environ.get("GSR_DIST_CHECK_N", "300000"))  # 38 tile rows, the last one partial
    cols = synthetic.mip360_like(n, 77)
    scene = renderer.GaussianScene.from_columns(cols, device=dev)
    fx = synthetic.pinhole_focal(W)
    cams = [renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)[:5]]
    plan = gdist.TileRowPlan(H, W, world)
    slots = int(os.environ.get("GSR_DIST_CHECK_SLOTS", "3"))
    views = int(os.environ.get("GSR_DIST_CHECK_VIEWS", "1"))  # frames per launch sequence and per gather (bench.py --views-per-launch)
    fif = renderer.FramesInFlight(scene, slots=slots, views=views)
    opts = renderer.make_options(**plan.shard_options(rank))
    fif.set_max_pairs(max(fif.rasterizers[0].fit_pairs(c, opts) for c in cams))
    torch.cuda.synchronize(dev)
    # bench.py's schedule: `slots` frames (batches of `views` frames) in flight, each on its own stream / workspace / wire buffer,
    # gathers asynchronous
    seq = cams + cams[:2]  # 7 frames: every buffer is reused at least once
    frames = []
    if views == 1:
        sf = gdist.ShardedFrames(plan, rank, dev, slots, lambda k, c, strip: fif.rasterizers[k].enqueue(c, opts, out=strip),
                                 streams=fif.streams)
        subs, sizes = seq, [1] * len(seq)
    else:
        sf = gdist.ShardedFrames(plan, rank, dev, slots, lambda k, cs, strips: fif.rasterizers[k].enqueue_batch(cs, opts, out=strips[: len(cs)]),
                                 streams=fif.streams, views=views)
        subs = [seq[i: i + views] for i in range(0, len(seq), views)]   # the last batch is partial
        sizes = [len(b) for b in subs]
    done = 0

    def keep(f):
        nonlocal done
        if rank == 0 and f is not None:
            frames.extend(f[: sizes[done]].clone().unbind(0) if views > 1 else [f.clone()])
        if f is not None or rank != 0:
            done += 1

    for c in subs:
        f = sf.submit(c)
        if f is not None:
            keep(f)
    while sf.pending:
        keep(sf._finish_oldest())
    for k in range(slots):
        fif.stats(k)
    cams = seq
    ok = True
    if rank == 0:
        full = renderer.Rasterizer(scene)
        for i, c in enumerate(cams):
            ref = full.render(c)
            if not torch.equal(ref, frames[i]):
                ok = False
                print(f"frame {i}: gathered frame differs from the unsharded render, max abs {(ref - frames[i]).abs().max().item():.3e}", flush=True)
        assert frames[0].any() and not torch.equal(frames[0], frames[1])
    flag = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu")
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and ok:
        print(f"DIST_CHECK_OK backend={backend} ranks={world} frames={len(cams)}", flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
