#!/usr/bin/env python3
"""Throughput experiment: stage pipelining instead of frame slots.  Stream A runs preprocess + bin/sort of frame f + 1 while
stream B blends frame f (one blend at a time, always next to HBM/latency-bound work), over W workspaces.
With GSR_CU_SPLIT=a (1..7) the two streams are created with complementary CU masks (hipExtStreamCreateWithCUMask): stream A
gets a of every 8 CUs, stream B the rest — spatial partitioning instead of hoping for co-scheduling.
usage: stage_pipeline.py [W ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsr_amd
from gsr_amd import renderer, synthetic
from gsr_amd._lib import check, lib

n, W_, H = 6_131_954, 1920, 1080
cols = synthetic.mip360_like(n, 361)
p = synthetic.ring_cameras(25)[0]
fx = synthetic.pinhole_focal(W_)
cam = renderer.make_camera(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W_, 2 * H, W_, H)
scene = renderer.GaussianScene.from_columns(cols, sh_half=os.environ.get("GSR_SH_HALF") == "1")
opts = renderer.make_options()
R0 = renderer.Rasterizer(scene)
mp = R0.fit_pairs(cam, opts)
sc = scene.c_struct()
for NW in [int(x) for x in (sys.argv[1:] or ["2", "3", "4"])]:
    Rs = [renderer.Rasterizer(scene, max_pairs=mp) for _ in range(NW)]
    wss = [r._workspace(W_, H) for r in Rs]
    outs = [torch.zeros((H, W_, 3), device="cuda") for _ in range(NW)]
    split = int(os.environ.get("GSR_CU_SPLIT", "0"))
    if split:
        hip = C.CDLL("libamdhip64.so")  # already loaded by torch: the same runtime
        def masked(keep):
            words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if keep((32 * w + b) % 8)) for w in range(8)])
            st = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
            assert rc == 0, rc
            return torch.cuda.ExternalStream(st.value)
        A, B = masked(lambda r: r < split), masked(lambda r: r >= split)
    else:
        # GSR_PRIO=1: stream A (preprocess + sorts: HBM- and latency-bound, short workgroups) at HIGH priority, the blend at normal
        # priority: A's workgroups take the slots the blend's workgroups free, the blend's resident ones keep computing
        prio = int(os.environ.get("GSR_PRIO", "0"))
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        A = torch.cuda.Stream(priority=-1 if prio == 1 else (0 if prio == 0 else 0))
        B = torch.cuda.Stream(priority=-1 if prio == 2 else 0)
    sorted_ev = [torch.cuda.Event() for _ in range(NW)]
    blended_ev = [torch.cuda.Event() for _ in range(NW)]
    torch.cuda.synchronize()

    split_after_pre = os.environ.get("GSR_SPLIT") == "pre"  # stream A: preprocess only; stream B: bin/sort + blend

    def run(K):
        if split_after_pre:
            for f in range(K):
                k = f % NW
                ws = wss[k]
                if f >= NW:
                    A.wait_event(blended_ev[k])
                check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), None, A.cuda_stream))
                sorted_ev[k].record(A)
                B.wait_event(sorted_ev[k])
                check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), mp, ws.data_ptr(), ws.numel(), B.cuda_stream))
                check(lib.gsr_blend(None, n, C.byref(cam), C.byref(opts), mp, ws.data_ptr(), ws.numel(), outs[k].data_ptr(), None, B.cuda_stream))
                blended_ev[k].record(B)
            return
        for f in range(K):
            k = f % NW
            ws = wss[k]
            if f >= NW:
                A.wait_event(blended_ev[k])      # the workspace's previous frame has been blended
            check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), None, A.cuda_stream))
            check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), mp, ws.data_ptr(), ws.numel(), A.cuda_stream))
            sorted_ev[k].record(A)
            B.wait_event(sorted_ev[k])
            check(lib.gsr_blend(None, n, C.byref(cam), C.byref(opts), mp, ws.data_ptr(), ws.numel(), outs[k].data_ptr(), None, B.cuda_stream))
            blended_ev[k].record(B)

    run(2 * NW + 2); torch.cuda.synchronize()
    K = 60
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ref = R0.render(cam, opts)
    print(f"stage pipeline, split {os.environ.get('GSR_SPLIT', 'pre+sort|blend')}, sh_half {scene.sh_half}, prio {os.environ.get('GSR_PRIO', '0')}, CU split {split}/8, {NW} workspaces: {dt * 1e3:.3f} ms per frame ({1 / dt:.0f} frames/s)  frames identical: {all(torch.equal(o, ref) for o in outs)}", flush=True)
