#!/usr/bin/env python3
"""Experiment: batches in flight on streams that own DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) instead of sharing the whole
machine — do a blend on one half and the HBM-bound stages on the other half co-run better than kernels that each fill every CU?
usage: cu_mask_experiment.py [K] [frames]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd  # noqa: F401
from gsr_amd import renderer, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 96
W, H = 1920, 1080
cols = synthetic.mip360_like(6_131_954, 361)
fx = synthetic.pinhole_focal(W)
ring = [renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)]
scene = renderer.GaussianScene.from_columns(cols)
del cols
opts0 = renderer.make_options()
probe = renderer.Rasterizer(scene)
need = max(probe.fit_pairs(c, opts0) for c in ring)
passes = probe.sort_passes
opts = probe.bounded(opts0)
del probe
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return torch.cuda.ExternalStream(st.value)


def run(streams, S, frames, cams):
    fif = renderer.FramesInFlight(scene, slots=S, max_pairs=need, views=K)
    fif.set_sort_passes(passes)
    if streams is not None:
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        fif.streams = list(streams)
    outs = [torch.zeros((K, H, W, 3), device="cuda") for _ in range(S)]
    nb = (frames + K - 1) // K

    def loop(count):
        for b in range(count):
            cs = [cams[(b * K + j) % len(cams)] for j in range(K)]
            fif.submit_batch(cs, opts, out=outs[b % S], slot=b % S)

    loop(2 * S + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(nb)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for k in range(S):
        fif.stats(k)
    return nb * K / el


sets = {
    "shared (torch streams)": None,
    "halves 0-127 | 128-255": [set(range(0, 128)), set(range(128, 256))],
    "even | odd CUs": [set(range(0, 256, 2)), set(range(1, 256, 2))],
    "interleaved by 8: XCD-like": [set(i for i in range(256) if (i % 8) < 4), set(i for i in range(256) if (i % 8) >= 4)],
    "three quarters | quarter": [set(range(0, 192)), set(range(192, 256))],
    "even | odd, two streams each": [set(range(0, 256, 2)), set(range(1, 256, 2)), set(range(0, 256, 2)), set(range(1, 256, 2))],
    "thirds (CU % 3)": [set(i for i in range(256) if i % 3 == r) for r in range(3)],
    "4 of 8 | other 4, two streams each": [set(i for i in range(256) if (i % 8) < 4), set(i for i in range(256) if (i % 8) >= 4)] * 2,
}
for label, cams in (("camera set", ring), ("one camera", ring[:1])):
    print(label, flush=True)
    for name, masks in sets.items():
        try:
            streams = None if masks is None else [masked_stream(m) for m in masks]
            S = 2 if masks is None else len(masks)
            run(streams, S, 32, cams)
            print(f"  K={K} S={S} {name:36s} {run(streams, S, FRAMES, cams):7.1f} frames/s", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"  {name}: {e}", flush=True)
    # reference point: three shared streams (bench.py's default number of batches in flight)
    print(f"  K={K} S=3 shared (torch streams)            {run(None, 3, FRAMES, cams):7.1f} frames/s", flush=True)
