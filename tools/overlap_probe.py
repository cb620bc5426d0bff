#!/usr/bin/env python3
"""Can an HBM-bound kernel run UNDER the blend?  The blend of the bench frame on stream A next to a plain device copy of 0.7 GB
(1.4 GB of traffic: what the preprocess moves) on stream B: alone, alone, together.  torch's copy kernel is small-footprint
(few VGPRs, no LDS), i.e. the best case for finding room beside the blend's resident workgroups."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import gsr_amd  # noqa: F401
from gsr_amd import renderer, utils
from gsr_amd._lib import check, lib

class A: workload = "bicycle"; gaussians = 0; camera = 0; camera_set = "single"
dev = torch.device("cuda", 0)
cols, cam_list, n, W, H, _ = bench.build_workload("bicycle", A, 0)
scene = renderer.GaussianScene.from_packed(utils.pack_gaussians(cols), device=dev, sh_half=os.environ.get("GSR_PROBE_SH_HALF") == "1")
print("sh_half", scene.sh_half)
del cols
cam = renderer.make_camera(*cam_list[0])
R = renderer.Rasterizer(scene)
R.fit_pairs(cam)
opts = R.bounded()
ws = R._workspace(W, H)
sc = scene.c_struct()
out = torch.empty((H, W, 3), device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), None, sa.cuda_stream))
check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), sa.cuda_stream))
src = torch.empty(700_000_000 // 4, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()

def blend(k):
    for _ in range(k):
        check(lib.gsr_blend(None, n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), out.data_ptr(), None, sa.cuda_stream))
def copy(k):
    with torch.cuda.stream(sb):
        for _ in range(k):
            dst.copy_(src)
def timed(f, reps=20):
    f(3); torch.cuda.synchronize(); t0 = time.perf_counter(); f(reps); torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
# synthetic HBM-bound kernels (tools/probe_kernels.hip): which property lets a kernel run under the blend?
probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so")
if os.path.exists(probe):
    P = C.CDLL(probe)
    P.probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    nvec = src.numel() // 4
    idx = torch.randperm(nvec, device=dev, dtype=torch.int32)
    idx = (torch.arange(nvec, device=dev, dtype=torch.int64) // 64 * 64 + (idx.long() % 64)).clamp_(max=nvec - 1).int()  # shuffled within 1-KB blocks: coalesced-ish gather
    torch.cuda.synchronize()
    tb0 = timed(blend)
    for v, name in ((0, "one float4 per thread (short-lived waves)"), (5, "4 serialised float4 per thread"), (1, "32 serialised float4 per thread (long-lived waves)"),
                    (2, "16 float4 per thread in flight (fat waves, 64+ VGPRs)"), (3, "index load -> gather (two dependent round trips)"),
                    (4, "load -> 600 FMAs -> store (compute tail)")):
        def run(k, v=v):
            for _ in range(k):
                assert P.probe_launch(v, src.data_ptr(), dst.data_ptr(), idx.data_ptr(), nvec, sb.cuda_stream) == 0
        t1 = timed(run)
        def both3(k, run=run):
            for _ in range(k):
                blend(1); run(1)
        t2 = timed(both3)
        print(f"probe kernel: {name:58s} alone {t1:.3f} ms, beside the blend ({tb0:.3f}) {t2:.3f} ms -> hidden {100 * (tb0 + t1 - t2) / t1:.0f} % of its time", flush=True)
R2 = renderer.Rasterizer(scene, max_pairs=R.max_pairs)
ws2 = R2._workspace(W, H)
def pre(k):
    for _ in range(k):
        check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws2.data_ptr(), ws2.numel(), None, sb.cuda_stream))
def presort(k):
    for _ in range(k):
        check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws2.data_ptr(), ws2.numel(), None, sb.cuda_stream))
        check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), R.max_pairs, ws2.data_ptr(), ws2.numel(), sb.cuda_stream))
def sort_only(k):
    for _ in range(k):
        check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), R.max_pairs, ws2.data_ptr(), ws2.numel(), sb.cuda_stream))
tb = timed(blend); tc = timed(copy)
for name, fn in (("preprocess", pre), ("preprocess + bin/sort", presort), ("bin/sort", sort_only)):
    if name == "bin/sort":
        pre(1); torch.cuda.synchronize()
    t1 = timed(fn)
    def both2(k, fn=fn):
        for _ in range(k):
            blend(1); fn(1)
    t2 = timed(both2)
    print(f"blend alone {tb:.3f} ms, {name} alone {t1:.3f} ms, one of each per iteration on two streams {t2:.3f} ms (sum {tb + t1:.3f}, max {max(tb, t1):.3f})", flush=True)
def both(k):
    for _ in range(k):
        blend(1); copy(1)
tbc = timed(both)
print(f"blend alone {tb:.3f} ms, 0.7 GB copy alone {tc:.3f} ms, one of each per iteration on two streams {tbc:.3f} ms "
      f"(sum {tb + tc:.3f}, max {max(tb, tc):.3f})")
