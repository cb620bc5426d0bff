#!/usr/bin/env python3
"""Where does the blend kernel's time go?  Times gsr_blend alone (lists already built) under settings that remove
one cost at a time: thresholded early-out (fewer evaluations), early_out_T = 2 (every wave stops after its first
64 entries: staging + launch + store only), draw_limit = 1 (near-empty lists: launch + store only).  GPU analysis tool.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="garden")
    ap.add_argument("--gaussians", type=int, default=0)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--ab-rounds", type=int, default=0, help="only compare the two blend implementations, interleaved")
    a = ap.parse_args()

    import bench
    from gsr_amd import renderer
    from gsr_amd._lib import check, lib

    args = argparse.Namespace(workload=a.workload, gaussians=a.gaussians, camera=0, camera_set="one",
                              input_dir=None, trained_model_path=None)
    cols, cam_list, n, W, H, _ = bench.build_workload(args.workload, args, getattr(args, "gaussians", 0))
    dev = torch.device("cuda:0")
    scene = renderer.GaussianScene.from_columns(cols, device=dev)
    cam = renderer.make_camera(*cam_list[0])
    R = renderer.Rasterizer(scene)
    R.render(cam)  # sizes the pair buffer
    ws = R._workspace(W, H)
    sc = scene.c_struct()
    out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    sp = int(torch.cuda.current_stream(dev).cuda_stream)

    def time_blend(label, build_opts, blend_opts):
        check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(build_opts), ws.data_ptr(), ws.numel(), None, sp))
        check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(build_opts), R.max_pairs, ws.data_ptr(), ws.numel(), sp))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for i in range(a.reps + 3):
            e = ev[max(i - 3, 0)]
            e[0].record()
            check(lib.gsr_blend(None, n, C.byref(cam), C.byref(blend_opts), R.max_pairs, ws.data_ptr(), ws.numel(), out.data_ptr(), None, sp))
            e[1].record()
        torch.cuda.synchronize()
        ms = float(np.median([x.elapsed_time(y) for x, y in ev]))
        st = R.stats()
        print(f"{label:34s} blend {ms * 1e3:8.1f} us   evaluated/launch {st['wave_entries'] / 1e6:7.2f} M   "
              f"fetched/launch {st['fetched_entries'] / 1e6:7.2f} M   pairs {st['n_pairs'] / 1e6:6.2f} M", flush=True)

    mk = renderer.make_options
    if a.ab_rounds:
        for _ in range(a.ab_rounds):
            time_blend("exact, valu", mk(), mk())
            time_blend("early_out_T=1e-4, valu", mk(), mk(early_out_T=1e-4))
        return
    time_blend("exact", mk(), mk())
    for T in (1e-6, 1e-4, 1e-2, 0.5):
        time_blend(f"early_out_T={T:g}", mk(), mk(early_out_T=T))
    time_blend("early_out_T=2 (first chunk only)", mk(), mk(early_out_T=2.0))
    time_blend("draw_limit=1 (near-empty lists)", mk(draw_limit=1), mk(draw_limit=1))


if __name__ == "__main__":
    main()
