#!/usr/bin/env python3
"""Phase timeline of one radix scatter launch of the tile sort: the final pass (sort.hip built with -DGSR_SORT_TRACE) or
pass 0, the one that drops culled pairs (-DGSR_SORT_TRACE -DGSR_SORT_TRACE_DROP=1, run with GSR_TRACE_PASS0=1).  Phases: 0 start, 1 keys loaded, 2 ranked, 3 bases scanned, 4 reordered in LDS, 5 written out.
GPU analysis tool."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from gsr_amd import renderer
    from gsr_amd._lib import lib

    args = argparse.Namespace(workload="garden", gaussians=0, camera=0, camera_set="one", input_dir=None, trained_model_path=None)
    cols, cam_list, n, W, H, _ = bench.build_workload(args)
    dev = torch.device("cuda:0")
    scene = renderer.GaussianScene.from_columns(cols, device=dev)
    cam = renderer.make_camera(*cam_list[0])
    R = renderer.Rasterizer(scene)
    for _ in range(3):
        R.render(cam)
    torch.cuda.synchronize()
    buf = np.zeros(16384 * 8, np.uint32)
    raw = C.CDLL(None)
    rc = lib.gsr_debug_sort_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
    assert rc == 0, rc
    st = buf.reshape(-1, 8)
    nblk = ((R.last_stats["n_pairs_bbox"] if os.environ.get("GSR_TRACE_PASS0") else R.last_stats["n_pairs"]) + 4095) // 4096
    st = st[:nblk].astype(np.int64)
    base = st[:, 0].min()
    t = (st[:, :6] - base) * 0.01
    print(f"workgroups {nblk}  span {t[:, 5].max():.1f} us")
    names = ["load", "rank", "scan+bases", "LDS reorder", "write-out"]
    d = np.diff(t, axis=1)
    for k, nm in enumerate(names):
        print(f"  {nm:12s} mean {d[:, k].mean():6.2f} us  p50 {np.median(d[:, k]):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}")
    tot = t[:, 5] - t[:, 0]
    print(f"  workgroup    mean {tot.mean():6.2f} us  p90 {np.percentile(tot, 90):6.2f}; mean concurrency {tot.sum() / t[:, 5].max():.0f}")
    edges = np.linspace(0, t[:, 5].max(), 11)
    for lo, hi in zip(edges[:-1], edges[1:]):
        c = 0.5 * (lo + hi)
        print(f"   {lo:6.1f}-{hi:6.1f} us resident {int(((t[:, 0] <= c) & (t[:, 5] > c)).sum())}")


if __name__ == "__main__":
    main()
