#!/usr/bin/env python3
"""How much does block-level culling (GsrScene.block_bounds) skip on the bench scene, and what does it do to the preprocess?
Per ring camera: fraction of the 256-gaussian blocks skipped for the whole frame and for rank 3 of 8; preprocess time with / without."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import gsr_amd  # noqa: F401
from gsr_amd import renderer, synthetic, dist as gdist
from gsr_amd._lib import check, lib

W, H = 1920, 1080
cols = synthetic.mip360_like(6_131_954, 361)
fx = synthetic.pinhole_focal(W)
ring = [renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)]
for spatial in (True, False):
    scene = renderer.GaussianScene.from_columns(cols, spatial_order=spatial)
    bare = renderer.GaussianScene({k: scene.t[k] for k in scene.FIELDS})
    o8 = renderer.make_options(**gdist.TileRowPlan(H, W, 8).shard_options(3))
    fr = [float(scene.blocks_skipped(c).float().mean()) for c in ring[::4]]
    fr8 = [float(scene.blocks_skipped(c, o8).float().mean()) for c in ring[::4]]
    print(f"{'morton' if spatial else 'file'} order: blocks skipped, whole frame {np.mean(fr):.3f} ({min(fr):.3f}..{max(fr):.3f}); rank 3 of 8 {np.mean(fr8):.3f}", flush=True)
    for name, sc_, in (("with bounds", scene), ("without", bare)):
        for oname, o in (("frame", renderer.make_options()), ("rank 3 of 8", o8)):
            R = renderer.Rasterizer(sc_)
            ws = R._workspace(W, H)
            s = sc_.c_struct()
            sp = int(torch.cuda.current_stream().cuda_stream)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(25):
                if i == 5:
                    ev[0].record()
                check(lib.gsr_preprocess(C.byref(s), C.byref(ring[0]), C.byref(o), ws.data_ptr(), ws.numel(), None, sp))
            ev[1].record()
            ev[1].synchronize()
            print(f"   preprocess {oname:12s} {name:12s}: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:7.1f} us", flush=True)
