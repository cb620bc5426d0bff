#!/usr/bin/env python3
"""Throughput experiment: frames of independent views in flight on S HIP streams (one workspace each), so that one frame's
HBM-bound stages overlap another's VALU-bound blend.  usage: stream_overlap.py [workload] [G r]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd
from gsr_amd import renderer, synthetic, dist as gdist

n = 6_131_954
W, H = 1920, 1080
cols = synthetic.mip360_like(n, 361)
cams = synthetic.ring_cameras(25)
fx = synthetic.pinhole_focal(W)
cam = renderer.make_camera(cams[0].qvec, cams[0].tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
scene = renderer.GaussianScene.from_columns(cols)
shard = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else None
if shard:
    plan = gdist.TileRowPlan(H, W, shard[0])
    opts = renderer.make_options(**plan.shard_options(shard[1]))
    shape = plan.strip_shape(shard[1])
else:
    opts, shape = renderer.make_options(), (H, W, 3)
for S in [int(x) for x in os.environ.get("GSR_OVERLAP_SLOTS", "1,2,3,4").split(",")]:
    Rs = [renderer.Rasterizer(scene) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [torch.zeros(shape, device="cuda") for _ in range(S)]
    for R in Rs:
        R.fit_pairs(cam, opts)  # also learns the depth-sort bound
    torch.cuda.synchronize()
    def run(k):
        for f in range(k):
            with torch.cuda.stream(streams[f % S]):
                Rs[f % S].enqueue(cam, opts, outs[f % S])
    run(2 * S); torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ok = all(torch.equal(o, outs[0]) for o in outs)
    print(f"streams {S}: {dt * 1e3:.3f} ms per frame ({1 / dt:.0f} frames/s)  frames identical: {ok}", flush=True)
