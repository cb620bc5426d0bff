#!/usr/bin/env python3
"""One-GPU rehearsal of the tile-row sharding: time each rank's shard of the bench frame separately, with one frame in flight
(per-frame latency) and with GSR_VIEWS frames per launch sequence x GSR_SLOTS batches in flight on separate streams (throughput,
what bench.py --gpus N runs).
The slowest shard (+ the framebuffer gather, not measured here) bounds the G-GPU frame time.
usage: shard_timing.py [bicycle|garden] [G r]      ("G r": only that shard, e.g. under rocprofv3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd
from gsr_amd import renderer, synthetic, dist as gdist

args = sys.argv[1:]
workload = args.pop(0) if args and not args[0].isdigit() else "bicycle"
n, seed = {"bicycle": (6_131_954, 361), "garden": (5_834_784, 360)}[workload]
only = (int(args[0]), int(args[1])) if len(args) > 1 else None
SLOTS = int(os.environ.get("GSR_SLOTS", "3"))   # batches in flight (bench.py's default for tile-row shards)
VIEWS = int(os.environ.get("GSR_VIEWS", "4"))   # frames per launch sequence (gsr_render_batch)
W, H = 1920, 1080
cols = synthetic.mip360_like(n, seed)
p = synthetic.ring_cameras(25)[0]
fx = synthetic.pinhole_focal(W)
cam = renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
scene = renderer.GaussianScene.from_columns(cols, spatial_order=os.environ.get("GSR_MORTON", "1") == "1")  # the loaders' default; GSR_MORTON=0: file order


def timed(fif, opts, outs, frames=24, views=1):
    S = fif.slots

    def go(batches):
        for b in range(batches):
            if views == 1:
                fif.submit(cam, opts, out=outs[b % S][0], slot=b % S)
            else:
                fif.submit_batch([cam] * views, opts, out=outs[b % S], slot=b % S)

    go(2 * S + 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nb = (frames + views - 1) // views
    go(nb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (nb * views) * 1e3


for G in (1, 2, 4, 8) if only is None else (only[0],):
    plan = gdist.TileRowPlan(H, W, G, int(os.environ.get("GSR_BLOCK", "2")))
    one, many, vis, pairs = [], [], [], []
    for r in range(G) if only is None else (only[1],):
        impl = int(os.environ.get("GSR_BLEND_IMPL", "0"))  # A/B of blend kernels
        pre = int(os.environ.get("GSR_SHARD_PRE", "0"))  # GsrOptions.shard_preprocess: 0 auto, 1 whole-frame kernel, 2 three-phase kernel
        opts = renderer.make_options(blend_impl=impl, shard_preprocess=pre, **plan.shard_options(r)) if G > 1 else renderer.make_options(blend_impl=impl)
        shape = plan.strip_shape(r) if G > 1 else (H, W, 3)
        fif = renderer.FramesInFlight(scene, slots=SLOTS, views=VIEWS)
        fif.set_max_pairs(fif.rasterizers[0].fit_pairs(cam, opts))
        outs = [torch.zeros((VIEWS,) + tuple(shape), device="cuda") for _ in range(SLOTS)]
        fif.set_sort_passes(fif.rasterizers[0].sort_passes)  # the depth-sort bound learned by the probing frame
        opts = fif.rasterizers[0].bounded(opts)
        single = renderer.FramesInFlight(scene, slots=1, max_pairs=fif.rasterizers[0].max_pairs)
        single.set_sort_passes(fif.rasterizers[0].sort_passes)
        timed(single, opts, outs[:1], 6)  # warm the clocks before the first measurement of the process
        one.append(timed(single, opts, outs[:1]))
        many.append(timed(fif, opts, outs, 48, VIEWS))
        st = fif.stats(0); vis.append(st["n_visible"]); pairs.append(st["n_pairs"])
        del fif, single, outs
    print(f"{workload} G={G}: shard ms, one frame in flight: min {min(one):.3f} max {max(one):.3f} | {VIEWS} per launch sequence x {SLOTS} in flight: min {min(many):.3f} "
          f"max {max(many):.3f} | visible/shard {min(vis)}..{max(vis)}  list entries/shard {min(pairs)}..{max(pairs)}", flush=True)
    print("   per rank, in flight: " + " ".join(f"{x:.3f}" for x in many) + " | one frame: " + " ".join(f"{x:.3f}" for x in one), flush=True)
