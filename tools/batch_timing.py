#!/usr/bin/env python3
"""Throughput of the bench scene over the 25-camera ring, a different camera every frame, as a function of the views per launch
sequence (K: Rasterizer(views=K), gsr_render_batch) and the batches in flight (S: FramesInFlight slots) — and the same for ONE
repeated camera.  Every configuration renders frames that are bit-identical to single-view renders (tests/test_gpu_parity.py).
usage: batch_timing.py [bicycle|garden] [K,K,...] [S,S,...] [frames]        GSR_SHARD="G r": time rank r's shard of G instead"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd  # noqa: F401
from gsr_amd import renderer, synthetic, dist as gdist

args = sys.argv[1:]
workload = args.pop(0) if args and not args[0][0].isdigit() else "bicycle"
Ks = [int(x) for x in args[0].split(",")] if len(args) > 0 else [1, 2, 4, 8]
Ss = [int(x) for x in args[1].split(",")] if len(args) > 1 else [1, 2, 3, 6]
FRAMES = int(args[2]) if len(args) > 2 else 96
n, seed = {"bicycle": (6_131_954, 361), "garden": (5_834_784, 360)}[workload]
W, H = 1920, 1080
cols = synthetic.mip360_like(n, seed)
fx = synthetic.pinhole_focal(W)
ring = [renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)]
scene = renderer.GaussianScene.from_columns(cols)
del cols
shard = os.environ.get("GSR_SHARD")
kw = {}
shape = (H, W, 3)
if shard:
    G, r = (int(x) for x in shard.split())
    plan = gdist.TileRowPlan(H, W, G, int(os.environ.get("GSR_BLOCK", "2")))
    kw = plan.shard_options(r)
    shape = plan.strip_shape(r)
opts0 = renderer.make_options(**kw)
probe = renderer.Rasterizer(scene)
need = max(probe.fit_pairs(c, opts0) for c in ring)
passes = probe.sort_passes
opts = probe.bounded(opts0)
del probe
print(f"{workload}{' shard ' + shard if shard else ''}: max_pairs {need}, depth-sort passes {passes}", flush=True)


def run(cams, K, S, frames):
    fif = renderer.FramesInFlight(scene, slots=S, max_pairs=need, views=K)
    fif.set_sort_passes(passes)
    outs = [torch.zeros((K,) + shape, device="cuda") for _ in range(S)]
    nb = (frames + K - 1) // K

    def loop(count):
        for b in range(count):
            cs = [cams[(b * K + j) % len(cams)] for j in range(K)]
            if K == 1:
                fif.submit(cs[0], opts, out=outs[b % S][0], slot=b % S)
            else:
                fif.submit_batch(cs, opts, out=outs[b % S], slot=b % S)

    loop(2 * S + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(nb)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for k in range(S):
        fif.stats(k)  # raises if any frame exceeded a bound
    return nb * K / el


for label, cams in (("camera set (25, a different one every frame)", ring), ("one camera repeated", ring[:1])):
    print(label, flush=True)
    for K in Ks:
        row = []
        for S in Ss:
            run(cams, K, S, max(K * S * 2, 16))  # warm
            row.append(f"S={S}: {run(cams, K, S, FRAMES):7.1f}")
        print(f"  K={K}  frames/s  " + "   ".join(row), flush=True)
