#!/usr/bin/env python3
"""One-GPU rehearsal of the tile-row sharding: time each rank's shard of the bench frame separately.
The slowest shard (+ the framebuffer gather, not measured here) bounds the G-GPU frame time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd
from gsr_amd import renderer, synthetic, utils, dist as gdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_834_784
only = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None   # "G r": just this shard (for a rocprofv3 timeline)
W, H = 1920, 1080
cols = synthetic.mip360_like(n, 360)
p = synthetic.ring_cameras(25)[0]
fx = synthetic.pinhole_focal(W)
cam = renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
scene = renderer.GaussianScene.from_columns(cols)
for G in (1, 2, 4, 8) if only is None else (only[0],):
    plan = gdist.TileRowPlan(H, W, G)
    times, vis, pairs = [], [], []
    for r in range(G) if only is None else (only[1],):
        R = renderer.Rasterizer(scene)
        opts = renderer.make_options(**plan.shard_options(r)) if G > 1 else renderer.make_options()
        R.fit_pairs(cam, opts)
        out = torch.zeros(plan.strip_shape(r) if G > 1 else (H, W, 3), device="cuda")
        for _ in range(3): R.enqueue(cam, opts, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): R.enqueue(cam, opts, out)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 20 * 1e3)
        st = R.stats(); vis.append(st["n_visible"]); pairs.append(st["n_pairs"])
    print(f"G={G}: shard ms min {min(times):.3f} max {max(times):.3f}  visible/shard {min(vis)}..{max(vis)}  pairs/shard {min(pairs)}..{max(pairs)}  "
          f"=> speedup bound {times[0] if G == 1 else 0:.3f}" if G == 1 else
          f"G={G}: shard ms min {min(times):.3f} max {max(times):.3f}  visible/shard {min(vis)}..{max(vis)}  pairs/shard {min(pairs)}..{max(pairs)}")
