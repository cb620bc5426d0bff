#!/usr/bin/env python3
"""How full are the blend kernel's 64-lane evaluations?  CPU-side analysis (oracle preprocess + numpy).

For a random sample of the visible gaussians of a bench workload this counts, per gaussian,
  px   pixels that pass the reference's draw test (alpha > 1/255 and power <= 0, rasterize.py:291),
  q8   8x8 quadrants (one wave each in blend.hip) that contain at least one such pixel,
  q4   4x4 blocks that contain at least one,
  rect pixels of the clipped bounding box (what the reference's slice touches, rasterize.py:271-272).
px / (64*q8) is the best-case useful-lane fraction of the current wave-per-quadrant mapping; 16*q4 / (64*q8)
is how much work a 16-lane-granular mapping could at best keep.  Analysis tool only: imports oracle/.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden")
    ap.add_argument("--gaussians", type=int, default=0)
    ap.add_argument("--sample", type=int, default=20000)
    ap.add_argument("--camera", type=int, default=0)
    a = ap.parse_args()

    import bench
    from gsr_amd import utils
    from oracle import cpu_oracle

    args = argparse.Namespace(workload=a.workload, gaussians=a.gaussians, camera=a.camera, camera_set="one",
                              input_dir=None, trained_model_path=None)
    cols, cam_list, n, W, H, _ = bench.build_workload(args.workload, args, getattr(args, "gaussians", 0))
    packed = utils.pack_gaussians(cols)
    cam = cpu_oracle.camera(*cam_list[0])
    pre = cpu_oracle.preprocess(packed, cam)
    bb = pre["pixel_bboxes"]  # x_min, y_min, x_max, y_max (clipped; empty when culled)
    vis = np.flatnonzero((bb[:, 2] > bb[:, 0]) & (bb[:, 3] > bb[:, 1]))
    rng = np.random.default_rng(0)
    pick = rng.choice(vis, size=min(a.sample, len(vis)), replace=False)
    tot = dict(px=0, q8=0, q4=0, rect=0, t16=0)
    for i in pick:
        x0, y0, x1, y1 = (int(v) for v in bb[i])
        x1 = min(x1, W - 1)  # Q1: last column / row are never drawn
        y1 = min(y1, H - 1)
        if x1 <= x0 or y1 <= y0:
            continue
        xs = np.arange(x0, x1, dtype=np.float32)[:, None]
        ys = np.arange(y0, y1, dtype=np.float32)[None, :]
        mx, my = pre["screen_means"][i]
        A, B, Cc = pre["sigmas"][i]
        dx, dy = mx - xs, my - ys
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        alpha = np.minimum(np.float32(0.99), pre["opacity"][i] * np.exp(power))
        ok = (alpha > np.float32(1.0 / 255.0)) & (power <= 0)
        if not ok.any():
            tot["rect"] += ok.size
            continue
        gx, gy = np.nonzero(ok)
        gx += x0
        gy += y0
        tot["px"] += len(gx)
        tot["q8"] += len(np.unique((gx >> 3) * 4096 + (gy >> 3)))
        tot["q4"] += len(np.unique((gx >> 2) * 4096 + (gy >> 2)))
        tot["t16"] += len(np.unique((gx >> 4) * 4096 + (gy >> 4)))
        tot["rect"] += ok.size
    s = len(pick)
    scale = len(vis) / s
    print(f"visible {len(vis)}  sampled {s}")
    for k, v in tot.items():
        print(f"  {k:5s} per gaussian {v / s:9.2f}   frame estimate {v * scale / 1e6:9.1f} M")
    print(f"useful lanes, wave per 8x8 quadrant : {tot['px'] / (64.0 * tot['q8']):.3f}")
    print(f"useful lanes, 16 lanes per 4x4 block: {tot['px'] / (16.0 * tot['q4']):.3f}")
    print(f"work kept by 4x4 granularity        : {16.0 * tot['q4'] / (64.0 * tot['q8']):.3f}")
    print(f"quadrants per 16x16 tile entry      : {tot['q8'] / max(1, tot['t16']):.2f}")


if __name__ == "__main__":
    main()
