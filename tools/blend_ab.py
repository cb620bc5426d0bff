#!/usr/bin/env python3
"""A/B of blend kernel variants on the bench frame, interleaved rounds in ONE process (MI355X guide, rule 24):
per variant the blend stage alone (gsr_blend between two events) — median and min over rounds — plus bit-identity of the
frames and the counters.   usage: tools/blend_ab.py [--workload bicycle] [--impls 0,3] [--rounds 15] [--early-out-T 0]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import gsr_amd  # noqa: F401
from gsr_amd import renderer, utils
from gsr_amd._lib import check, lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bicycle")
    ap.add_argument("--gaussians", type=int, default=0)
    ap.add_argument("--impls", default="0,3")
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--early-out-T", type=float, default=0.0)
    ap.add_argument("--camera", type=int, default=0)
    ap.add_argument("--camera-set", default="single")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cols, cam_list, n, W, H, _ = bench.build_workload(a.workload, a, a.gaussians)
    scene = renderer.GaussianScene.from_packed(utils.pack_gaussians(cols), device=dev)
    del cols
    cam = renderer.make_camera(*cam_list[0])
    R = renderer.Rasterizer(scene)
    R.fit_pairs(cam)
    impls = [int(x) for x in a.impls.split(",")]
    ws = R._workspace(W, H)
    sc = scene.c_struct()
    stream = torch.cuda.current_stream(dev)
    sp = int(stream.cuda_stream)
    outs = {i: torch.empty((H, W, 3), dtype=torch.float32, device=dev) for i in impls}
    times = {i: [] for i in impls}
    stats = {}
    base = renderer.make_options(early_out_T=a.early_out_T)
    check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(base), ws.data_ptr(), ws.numel(), None, sp))
    check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(base), R.max_pairs, ws.data_ptr(), ws.numel(), sp))
    for rnd in range(a.rounds + 2):
        for i in impls:
            o = renderer.make_options(early_out_T=a.early_out_T, blend_impl=i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            check(lib.gsr_blend(None, n, C.byref(cam), C.byref(o), R.max_pairs, ws.data_ptr(), ws.numel(), outs[i].data_ptr(), None, sp))
            e1.record(stream)
            torch.cuda.synchronize(dev)
            if rnd >= 2:
                times[i].append(e0.elapsed_time(e1))
            if rnd == 0:
                stats[i] = R.stats()
    ref = impls[0]
    for i in impls:
        t = np.array(times[i])
        same = bool(torch.equal(outs[i], outs[ref]))
        d = float((outs[i] - outs[ref]).abs().max())
        print(f"impl {i}: blend median {np.median(t):.4f} ms  min {t.min():.4f} ms   frame == impl {ref}: {same} (max abs diff {d:.2e})  "
              f"wave_entries {stats[i]['wave_entries']} fetched {stats[i]['fetched_entries']}", flush=True)


if __name__ == "__main__":
    main()
