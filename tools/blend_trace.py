#!/usr/bin/env python3
"""Per-workgroup timeline of the blend kernel (needs libgsr.so built with -DGSR_BLEND_TRACE, see --help of the Makefile
comment in blend.hip): start/end of every tile's workgroup on the 100 MHz wall clock, read back from the blend_stats
slots.  Prints the concurrency profile and the longest workgroups.  GPU analysis tool.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden")
    ap.add_argument("--early-out-T", type=float, default=0.0)
    a = ap.parse_args()
    import bench
    from gsr_amd import renderer

    args = argparse.Namespace(workload=a.workload, gaussians=0, camera=0, camera_set="one", input_dir=None, trained_model_path=None)
    cols, cam_list, n, W, H, _ = bench.build_workload(args)
    dev = torch.device("cuda:0")
    scene = renderer.GaussianScene.from_columns(cols, device=dev)
    cam = renderer.make_camera(*cam_list[0])
    R = renderer.Rasterizer(scene)
    opts = renderer.make_options(early_out_T=a.early_out_T)
    for _ in range(3):
        R.render(cam, opts)
    ws = R._workspace(W, H)
    head = ws[:2048].cpu().numpy().view(np.uint32)
    off, slots = int(head[(40 + 1024) // 4]), int(head[(40 + 1024) // 4 + 1])
    st = ws[off:off + slots * 32].cpu().numpy().view(np.uint32).reshape(slots, 8)
    st = st[st[:, 7] > 0]
    t0 = st[:, 5].astype(np.int64)
    t1 = st[:, 6].astype(np.int64)
    base = t0.min()
    t0 = (t0 - base) * 0.01  # us
    t1 = (t1 - base) * 0.01
    dur = t1 - t0
    print(f"workgroups {len(st)}   kernel span {t1.max():.1f} us   sum of workgroup time {dur.sum() / 1e3:.1f} ms "
          f"-> mean concurrency {dur.sum() / t1.max():.0f} workgroups")
    ev = st[:, :4].sum(1)
    print(f"workgroup duration: mean {dur.mean():.1f}  p50 {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f} us")
    k = np.argsort(-dur)[:8]
    for i in k:
        print(f"   start {t0[i]:7.1f}  dur {dur[i]:7.1f} us  list {st[i, 7]:6d}  evaluated {ev[i]:6d}  fetched {st[i, 4]:6d}")
    edges = np.linspace(0, t1.max(), 21)
    print("concurrency over time (workgroups resident at the bin centre):")
    for lo, hi in zip(edges[:-1], edges[1:]):
        c = 0.5 * (lo + hi)
        print(f"   {lo:7.1f}-{hi:7.1f} us  {int(((t0 <= c) & (t1 > c)).sum()):5d}")
    # per-entry cost as a function of how loaded the chip is
    rate = dur / np.maximum(ev, 1)
    print(f"us per evaluated (quadrant, entry) per workgroup: p10 {np.percentile(rate, 10) * 1e3:.1f} ns  p50 {np.median(rate) * 1e3:.1f} ns  p90 {np.percentile(rate, 90) * 1e3:.1f} ns")


if __name__ == "__main__":
    main()
