#!/usr/bin/env python3
"""A few batches of K views of the bench scene on ONE stream, for rocprofv3 --kernel-trace (tools/trace_batch.sh).
usage: batch_profile.py K [batches] [camera-set: 0|1]        GSR_SHARD="G r": rank r's shard of G"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd  # noqa: F401
from gsr_amd import renderer, synthetic, dist as gdist

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
camset = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
W, H = 1920, 1080
cols = synthetic.mip360_like(6_131_954, 361)
fx = synthetic.pinhole_focal(W)
ring = [renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)]
if not camset:
    ring = ring[:1]
scene = renderer.GaussianScene.from_columns(cols)
del cols
kw, shape = {}, (H, W, 3)
if os.environ.get("GSR_SHARD"):
    G, r = (int(x) for x in os.environ["GSR_SHARD"].split())
    plan = gdist.TileRowPlan(H, W, G, int(os.environ.get("GSR_BLOCK", "2")))
    kw, shape = plan.shard_options(r), plan.strip_shape(r)
R = renderer.Rasterizer(scene, views=K)
o0 = renderer.make_options(**kw)
R.max_pairs = max(renderer.Rasterizer(scene).fit_pairs(c, o0) for c in ring[::6])
R.render(ring[0], o0)
opts = R.bounded(o0)
out = torch.zeros((K,) + shape, device="cuda")
for b in range(B):
    cams = [ring[(b * K + j) % len(ring)] for j in range(K)]
    if K == 1:
        R.enqueue(cams[0], opts, out=out[0])
    else:
        R.enqueue_batch(cams, opts, out=out)
torch.cuda.synchronize()
print(R.stats())
