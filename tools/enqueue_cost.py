#!/usr/bin/env python3
"""Host-side cost of enqueueing one frame (25 kernel launches through ctypes) against the GPU time of the frame: how far
ahead of the GPU the CPU runs.  GPU analysis tool."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from gsr_amd import renderer

    args = argparse.Namespace(workload="garden", gaussians=0, camera=0, camera_set="one", input_dir=None, trained_model_path=None)
    cols, cam_list, n, W, H, _ = bench.build_workload(args.workload, args, getattr(args, "gaussians", 0))
    scene = renderer.GaussianScene.from_columns(cols, device=torch.device("cuda:0"))
    cam = renderer.make_camera(*cam_list[0])
    R = renderer.Rasterizer(scene)
    R.fit_pairs(cam)
    out = torch.empty((H, W, 3), device="cuda")
    for shard in (1, 8):
        opts = renderer.make_options() if shard == 1 else renderer.make_options(tile_row_begin=0, tile_row_step=shard, output_layout=2)
        o = out if shard == 1 else None
        for _ in range(5):
            o = R.enqueue(cam, opts, out=o)
        torch.cuda.synchronize()
        k = 200
        t0 = time.perf_counter()
        for _ in range(k):
            R.enqueue(cam, opts, out=o)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"shard 1/{shard}: host enqueue {1e6 * (t1 - t0) / k:7.1f} us per frame; GPU drains the {k} frames in {1e6 * (t2 - t0) / k:7.1f} us per frame", flush=True)


if __name__ == "__main__":
    main()
