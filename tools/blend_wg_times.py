#!/usr/bin/env python3
"""Where does the blend kernel's time go across workgroups?  Needs a library built with -DGSR_BLEND_TIMESTAMPS
(tools/libgsr_ts.so, loaded through GSR_LIB_PATH): every blend workgroup leaves its start / end (100 MHz wall clock) and
its hardware id in its stats slot.  usage: GSR_LIB_PATH=tools/libgsr_ts.so python tools/blend_wg_times.py [G r]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gsr_amd
from gsr_amd import renderer, synthetic, dist as gdist

n, W, H = 6_131_954, 1920, 1080
cols = synthetic.mip360_like(n, 361)
p = synthetic.ring_cameras(25)[0]
fx = synthetic.pinhole_focal(W)
cam = renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
scene = renderer.GaussianScene.from_columns(cols)
shard = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else None
if shard:
    plan = gdist.TileRowPlan(H, W, shard[0])
    opts, shape = renderer.make_options(**plan.shard_options(shard[1])), plan.strip_shape(shard[1])
else:
    opts, shape = renderer.make_options(), (H, W, 3)
R = renderer.Rasterizer(scene)
R.fit_pairs(cam, opts)
out = torch.zeros(shape, device="cuda")
for _ in range(4):
    R.enqueue(cam, opts, out)
torch.cuda.synchronize()
ws = R._workspace(W, H).cpu().numpy()
off, slots = (int(x) for x in ws[2096:2104].view(np.uint32))      # FrameCtrl.stats_off / stats_slots (gsr_internal.h)
st = ws[off: off + slots * 32].view(np.uint32).reshape(slots, 8)
live = st[:, 7] != 0
t0 = st[live, 6].astype(np.int64); t1 = st[live, 7].astype(np.int64)
base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us
dur = t1 - t0
fetched = st[live, 4]
ev = st[live, 0:4].sum(1)
print(f"{live.sum()} workgroups, span {t1.max():.1f} us; start times: median {np.median(t0):.1f} us, last {t0.max():.1f} us")
print(f"workgroup duration: median {np.median(dur):.1f} us, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}")
for q in (50, 75, 90, 95, 99, 100):
    print(f"  {q:3d} % of the workgroups have ended by {np.percentile(t1, q):7.1f} us")
act = [(int(((t0 <= t) & (t1 > t)).sum())) for t in np.linspace(0, t1.max(), 11)]
print("workgroups in flight at 0,10,..,100 % of the span:", act)
k = np.argsort(-dur)[:8]
for i in k:
    print(f"  wg start {t0[i]:6.1f} end {t1[i]:6.1f} dur {dur[i]:6.1f} us  entries staged {fetched[i]:6d}  evaluated (4 waves) {ev[i]:6d}  ns per staged entry {1e3 * dur[i] / max(fetched[i], 1):.1f}")
# throughput view: evaluated entries per us per workgroup, long vs short lists
rate = ev / np.maximum(dur, 1e-3)
print(f"evaluated (quadrant, entry) per us per workgroup: median {np.median(rate):.1f}; of the 5 % longest lists {np.median(rate[fetched >= np.percentile(fetched, 95)]):.1f}")
