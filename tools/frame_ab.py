#!/usr/bin/env python3
"""A/B of whole-frame variants selected by GsrOptions fields (e.g. fine_binning=1, saturation_rule=1, colour_stage=1),
interleaved rounds in ONE process (MI355X guide, rule 24): per variant the three stage times (events around gsr_preprocess /
gsr_bin_sort / gsr_blend), median over rounds, plus bit-identity of the frames.
usage: tools/frame_ab.py [--workload bicycle] [--rounds 12] VARIANT [VARIANT ...]     VARIANT = name[:field=value[,field=value]]
(GSR_MORTON=0: the scene in file order instead of the loaders' Morton order)"""
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
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--camera", type=int, default=0)
    ap.add_argument("--camera-set", default="single")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cols, cam_list, n, W, H, _ = bench.build_workload(a.workload, a, a.gaussians)
    packed = utils.pack_gaussians(cols)
    del cols
    scene = renderer.GaussianScene.from_packed(packed, device=dev, spatial_order=os.environ.get("GSR_MORTON", "1") == "1")  # A/B of the loader option
    del packed
    cam = renderer.make_camera(*cam_list[0])
    variants = []
    for v in a.variants:
        name, _, envs = v.partition(":")
        kw = {k: (float(v) if "." in v or "e" in v else int(v)) for k, v in (e.split("=") for e in envs.split(",") if e)}
        variants.append((name, renderer.make_options(**kw)))

    R = renderer.Rasterizer(scene)
    need = 0
    for _, o in variants:  # one pair buffer that fits every variant
        need = max(need, R.fit_pairs(cam, o))
    R.max_pairs = need
    ws = R._workspace(W, H)
    sc = scene.c_struct()
    stream = torch.cuda.current_stream(dev)
    sp = int(stream.cuda_stream)
    outs = {name: torch.empty((H, W, 3), dtype=torch.float32, device=dev) for name, _ in variants}
    times = {name: [] for name, _ in variants}
    stats = {}
    for rnd in range(a.rounds + 2):
        for name, o in variants:
            opts = R.bounded(o)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for _ in range(2):  # the second repetition is the timed one (same variant back to back: caches warm for it)
                ev[0].record(stream)
                check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), None, sp))
                ev[1].record(stream)
                check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), sp))
                ev[2].record(stream)
                check(lib.gsr_blend(None, n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), outs[name].data_ptr(), None, sp))
                ev[3].record(stream)
            torch.cuda.synchronize(dev)
            if rnd >= 2:
                times[name].append([ev[k].elapsed_time(ev[k + 1]) for k in range(3)])
            if rnd == 0:
                stats[name] = R.stats()
    ref = variants[0][0]
    for name, _ in variants:
        t = np.median(np.array(times[name]), axis=0)
        print(f"{name:14s} preprocess {t[0]:.4f}  bin_sort {t[1]:.4f}  blend {t[2]:.4f}  frame {t.sum():.4f} ms   == {ref}: {bool(torch.equal(outs[name], outs[ref]))}  "
              f"D {stats[name]['n_pairs_bbox']} E {stats[name]['n_pairs']} fetched {stats[name]['fetched_entries']} evaluated {stats[name]['wave_entries']}", flush=True)


if __name__ == "__main__":
    main()
