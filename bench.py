#!/usr/bin/env python3
"""bench.py — frames/s of the MI355X forward rasterizer on BASELINE.json's metric workload.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

A step = one complete 1920x1080 frame of the synthetic stand-in for MipNeRF-360 'garden'
(BASELINE.json configs[1]; SURVEY.md §8(d): mip360_like(5_834_784, seed 360), ring camera 0), fp32,
reference_compat, no early termination: preprocess -> depth sort -> tile binning -> blend, scene resident
in HBM before the timed region.  With N > 1 the SAME frame is sharded by interleaved tile rows over the N
GPUs and gathered to rank 0 over RCCL (strong scaling: total work per frame fixed).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel (blend): algorithmic bytes 40*E + 12*P + 8*tiles per
                launch / its measured average duration (HIP events on the launch stream) vs 8 TB/s
  cpu_baseline  the reference's per-gaussian torch loop, ported (oracle/torch_loop.py), timed on this host on a
                bounded subsample of the same frame and extrapolated to the frame
and extras (PSNR of the timed configuration against the CPU oracle at full size, per-stage times, counters).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch

WORKLOADS = {
    # name: (generator, n, seed, W, H, description)
    "garden": ("mip360_like", 5_834_784, 360, 1920, 1080, "synthetic stand-in for MipNeRF-360 garden (configs[1])"),
    "bicycle": ("mip360_like", 6_131_954, 361, 1920, 1080, "synthetic stand-in for MipNeRF-360 bicycle (configs[2]): fp16 SH storage; "
                "blend accumulators stay fp32 (bf16 accumulators measure 41 dB < the 50 dB bar, SURVEY.md §7)"),
    "box4k": ("uniform_box", 20_000_000, 20, 3840, 2160, "20M uniform gaussians at 4K (configs[4])"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def host_threads() -> int:
    """CPU threads for the host-side legs: the reference's rule cpu_count()-1 (rasterize.py:323), applied to the
    CPUs this process may actually use, capped at the 16-CPU share a 1-GPU box is allotted."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 2
    share = int(os.environ.get("GSR_CPU_SHARE", "16"))
    return max(1, min(avail, share) - 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="garden", choices=sorted(WORKLOADS))
    ap.add_argument("--gaussians", type=int, default=0, help="override the gaussian count (0 = the workload's)")
    ap.add_argument("--early-out-T", type=float, default=0.0)
    ap.add_argument("--blend-impl", type=int, default=0, help="0/1 vector-ALU blend (reference-grade), 2 matrix-pipe blend")
    ap.add_argument("--overlap", action="store_true", help="two streams: run the SH colour pass under the sorts (measured slower)")
    ap.add_argument("--sh-half", action="store_true", help="store SH coefficients as fp16 (implied by --workload bicycle)")
    ap.add_argument("--camera", type=int, default=0)
    ap.add_argument("--camera-set", default="single", choices=["single", "all"],
                    help="single: every step renders --camera; all: steps cycle over the whole camera set (configs[3])")
    ap.add_argument("--input_dir", default="", help="real data: MipNeRF-360 scene dir with sparse/0/{images,cameras}.bin")
    ap.add_argument("--trained_model_path", default="", help="real data: INRIA model dir (point_cloud/iteration_30000/point_cloud.ply)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16-output", action="store_true", help="store the frame as bfloat16 (configs[2]); accumulation stays fp32")
    ap.add_argument("--no-early-out-leg", action="store_true", help="skip the extra T<1e-4 measurement (profiling runs)")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    return ap.parse_args()


def build_real_workload(args):
    """SURVEY.md §8(d) real-data mode: the trained ply + COLMAP poses; the frame is forced to 1920x1080 with a
    consistent pinhole (focal = fx_full * 1920 / cam.width); test cameras = every 8th image by sorted name."""
    from gsr_amd import ply, utils

    W, H = 1920, 1080
    images, cams = utils.read_scene(args.input_dir)
    cam0 = cams[1]
    f = float(cam0.params[0]) * W / float(cam0.width)
    ordered = sorted(images.values(), key=lambda im: im.name)
    test = ordered[::8]
    poses = test if args.camera_set == "all" else [ordered[args.camera % len(ordered)]]
    # cam_args in gsr_camera_setup's convention: full-res focal 2f over a 2W x 2H sensor -> fov and f/1 at W x H
    cam_list = [(im.qvec, im.tvec, 2 * f, 2 * f, 2 * W, 2 * H, W, H) for im in poses]
    cols = ply.read_gaussians_columns(os.path.join(args.trained_model_path, "point_cloud/iteration_30000/point_cloud.ply"))
    n = len(cols["x"])
    return cols, cam_list, n, W, H, f"real data: {args.input_dir} + {args.trained_model_path}, {len(cam_list)} camera(s)"


def build_workload(args):
    from gsr_amd import synthetic

    if args.input_dir and args.trained_model_path:
        return build_real_workload(args)
    gen, n, seed, W, H, desc = WORKLOADS[args.workload]
    if args.gaussians:
        n = args.gaussians
    cols = getattr(synthetic, gen)(n, seed)
    if gen == "uniform_box":
        poses = [synthetic.box_camera()]
    elif args.camera_set == "all":
        poses = synthetic.ring_cameras(25)
    else:
        poses = [synthetic.ring_cameras(25)[args.camera]]
    fx = synthetic.pinhole_focal(W)
    # on-disk convention: full-res = 2x, scale-factor 2
    cam_list = [(p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H) for p in poses]
    return cols, cam_list, n, W, H, desc


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # GSR_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # strips are staged through the host); the measured configuration is nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gsr_amd import dist as gdist
    from gsr_amd import renderer, utils

    cols, cam_list, n, W, H, desc = build_workload(args)
    cam_args = cam_list[0]
    packed = utils.pack_gaussians(cols)
    del cols
    sh_half = args.sh_half or args.workload == "bicycle"
    scene = renderer.GaussianScene.from_packed(packed, device=dev, sh_half=sh_half)
    cams = [renderer.make_camera(*c) for c in cam_list]
    cam = cams[0]
    ncam = len(cams)
    plan = gdist.TileRowPlan(H, W, world)
    out_dtype = torch.bfloat16 if args.bf16_output else torch.float32
    fg = gdist.FrameGather(plan, rank, dev, dtype=out_dtype)
    R = renderer.Rasterizer(scene, overlap=args.overlap)
    state1 = {"i": 0}
    if world == 1:  # no sharding: blend straight into the frame
        opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, output_bf16=args.bf16_output)
        strip_view = fg.frame

        def step():
            c = cams[state1["i"] % ncam]
            state1["i"] += 1
            return R.enqueue(c, opts, out=strip_view)
    else:
        opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, output_bf16=args.bf16_output,
                                     **plan.shard_options(rank))
        strip_view = fg.own_view(0)
        state = {"i": 0, "pending": None}

        def step():
            # frame k's strip is gathered asynchronously (RCCL stream) while frame k+1 renders into the other buffer
            buf = state["i"] & 1
            c = cams[state["i"] % ncam]
            state["i"] += 1
            R.enqueue(c, opts, out=fg.own_view(buf))
            h = fg.gather_async(buf)
            done = fg.finish(state["pending"]) if state["pending"] is not None else None
            state["pending"] = h
            return done

        def drain():
            out = fg.finish(state["pending"]) if state["pending"] is not None else None
            state["pending"] = None
            return out

    # size the pair buffer to the heaviest view once (grows on overflow), outside the timed region
    need = max(R.fit_pairs(c, opts) for c in cams)
    R.max_pairs = need
    R.render(cam, opts, out=strip_view)
    shard_stats = dict(R.last_stats)
    for _ in range(args.warmup):
        step()
    state1["i"] = 0
    if world > 1:
        drain()
        state["i"] = 0
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame = step()
    if world > 1:
        frame = drain()  # the last frame's gather completes inside the timed region
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    R.stats()  # raises if the last frame overflowed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        result = {
            "metric": "frames/sec @1080p + PSNR vs torch ref, MipNeRF-360 bicycle, 1/2/4/8 GPUs" if args.workload != "box4k"
                      else "frames/sec @4K, synthetic 20M gaussians",
            "value": args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "gaussians": n, "width": W, "height": H,
                       "camera": args.camera if ncam == 1 else f"cycling over {ncam} cameras",
                       "sharding": f"tile rows interleaved over {world} GPU(s), RCCL gather to rank 0" if world > 1 else "none",
                       "reference_compat": True, "early_out_T": args.early_out_T, "sh_storage": "f16" if sh_half else "f32",
                       "frame_storage": "bf16 (fp32 accumulation)" if args.bf16_output else "f32",
                       "blend_impl": "mfma" if args.blend_impl == 2 else "valu",
                       "streams": "2 (SH colour pass under the sorts)" if args.overlap else 1},
            "stats_rank0_shard": shard_stats,
        }

    # ---- rank 0: per-stage timing + roofline of ITS shard (the whole frame at N=1), outside the timed region; at N=1 also
    # ---- PSNR vs the oracle, the early-out leg and the CPU baseline -------------------------------------------------------
    if rank == 0:
        import ctypes as C

        if world == 1:
            frame = R.enqueue(cam, opts, out=strip_view)  # the frame checked against the oracle below is camera 0's

        from gsr_amd._lib import check, lib

        ws = R._workspace(W, H)
        sc = scene.c_struct()
        full_opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl,
                                          **(plan.shard_options(rank) if world > 1 else {}))
        out = torch.empty(plan.strip_shape(rank) if world > 1 else (H, W, 3), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev)
        sp = int(stream.cuda_stream)
        reps = max(10, min(50, args.steps))
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        for i in range(reps + 3):
            e = ev[max(i - 3, 0)]
            e[0].record(stream)
            check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(full_opts), ws.data_ptr(), ws.numel(), None, sp))
            e[1].record(stream)
            check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(full_opts), R.max_pairs, ws.data_ptr(), ws.numel(), sp))
            e[2].record(stream)
            check(lib.gsr_blend(n, C.byref(cam), C.byref(full_opts), R.max_pairs, ws.data_ptr(), ws.numel(), out.data_ptr(),
                                None, sp))
            e[3].record(stream)
        torch.cuda.synchronize(dev)
        st = R.stats()
        stage = [float(np.mean([e[k].elapsed_time(e[k + 1]) for e in ev])) for k in range(3)]
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        E, P, V = st["fetched_entries"], out.shape[0] * out.shape[1], st["n_visible"]  # E = entries actually fetched
        if world > 1:
            tiles = len(plan.rows[rank]) * ((W + 15) // 16)
        blend_bytes = 40.0 * E + 12.0 * P + 8.0 * tiles
        pre_bytes = (140.0 if sh_half else 236.0) * n + 64.0 * V
        achieved = blend_bytes / (stage[2] * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and world == 1:
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get("blend_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {"kernel": "gsr::blend_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "algorithmic_bytes_per_launch": blend_bytes, "avg_kernel_ms": stage[2],
                              "note": "blend is bound on-chip in exact mode (VALU issue + LDS broadcast pipe 72 % busy, DESIGN.md §5; SURVEY.md §7 hard part 1): see valu_issue_frac; traffic = rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE from profiles/"}
        # honesty figure: the blend is VALU-bound.  64 pixel evaluations per evaluated (quadrant, entry);
        # rocprofv3 PMC (profiles/r1_pmc.json) counts 21.4 VALU wave-instructions per of them (17 in the inner loop + culling,
        # staging).  Peak issue = one fp32 VALU wave-instruction per 2 cycles per SIMD (tools/valu_microbench.hip), 1024 SIMDs,
        # 2.4 GHz; v_exp_f32 costs 4 and v_cmp/v_cndmask ~1.5 of those slots, so the pipe is fuller than this fraction says.
        evals = 64.0 * st["wave_entries"]
        result["roofline"]["pixel_evaluations"] = evals
        result["roofline"]["valu_issue_frac"] = st["wave_entries"] * 21.4 / (stage[2] * 1e-3) / (1024 * 1.2e9)
        result["stage_ms"] = {"preprocess": stage[0], "bin_sort": stage[1], "blend": stage[2]}
        result["stage_hbm_gbs"] = {"preprocess": pre_bytes / (stage[0] * 1e-3) / 1e9}
        result["stats"] = st
        if world > 1:
            result["roofline"]["note"] = "rank 0's shard (interleaved tile rows); " + result["roofline"]["note"]

    if rank == 0 and world == 1:
        # the same frame with the usual 3DGS saturation cut-off (INRIA's T < 1e-4), timed the same way: the headline stays the
        # exact mode (no blend work skipped, reference semantics Q5); this line shows what north_star's "ballot early-out on
        # saturated alpha" buys inside its PSNR >= 50 dB tolerance
        eo_img = None
        if args.early_out_T == 0.0 and not args.no_early_out_leg:
            eo_T = 1e-4
            eo_opts = renderer.make_options(early_out_T=eo_T, blend_impl=args.blend_impl)
            eo_out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            for _ in range(min(args.warmup, 5) + 1):
                R.enqueue(cam, eo_opts, out=eo_out)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(args.steps):
                R.enqueue(cams[i % ncam], eo_opts, out=eo_out)
            torch.cuda.synchronize(dev)
            eo_elapsed = time.perf_counter() - t1
            R.enqueue(cam, eo_opts, out=eo_out)
            eo_stats = R.stats()
            eo_img = eo_out.cpu().numpy()
            result["early_out"] = {"early_out_T": eo_T, "frames_per_s": args.steps / eo_elapsed,
                                   "ms_per_step": 1e3 * eo_elapsed / args.steps, "wave_entries": eo_stats["wave_entries"],
                                   "fetched_entries": eo_stats["fetched_entries"],
                                   "note": "not the headline: same workload with the blend stopping a wave once T < 1e-4 for its 64 pixels"}

        oracle_img = None
        pre = order = None
        if not (args.no_psnr and args.no_cpu_baseline):
            from oracle import cpu_oracle as orc

            ocam = orc.camera(*cam_args)
            threads = host_threads()
            t1 = time.perf_counter()
            pre = orc.preprocess(packed, ocam)
            order = orc.depth_order(pre["cam_means"])
            t_pre = time.perf_counter() - t1
            if not args.no_psnr:
                t1 = time.perf_counter()
                screen, _, drawn = orc.composite(order, pre, W, H, threads=threads)
                t_comp = time.perf_counter() - t1
                oracle_img = screen.transpose(1, 0, 2)
                img = frame.float().cpu().numpy()
                mse = float(np.mean((img.astype(np.float64) - oracle_img) ** 2))
                result["psnr_vs_oracle_db"] = None if mse == 0 else 10 * np.log10(1.0 / mse)
                result["max_abs_vs_oracle"] = float(np.abs(img - oracle_img).max())
                if eo_img is not None:
                    mse = float(np.mean((eo_img.astype(np.float64) - oracle_img) ** 2))
                    result["early_out"]["psnr_vs_oracle_db"] = None if mse == 0 else 10 * np.log10(1.0 / mse)
                    result["early_out"]["max_abs_vs_oracle"] = float(np.abs(eo_img - oracle_img).max())
                result["cpu_oracle_c"] = {"frame_s": t_pre + t_comp, "threads": threads, "drawn": int(drawn),
                                          "note": "oracle/gsr_oracle.c, OpenMP over x bands; checker, not the baseline"}
        if not args.no_cpu_baseline:
            from oracle import torch_loop

            cores = host_threads()
            torch.set_num_threads(cores)  # the reference sets cpu_count()-1 (rasterize.py:323)
            s = torch_loop.timed_sample(pre, order, W, H, budget_s=args.cpu_budget_s, max_gaussians=400_000)
            result["cpu_baseline"] = {
                "value": 1.0 / s["extrapolated_frame_s"], "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": (f"reference per-gaussian torch loop (oracle/torch_loop.py): uniform random sample of {s['sampled']} of the "
                           f"frame's {s['total_iterations']} loop iterations, {s['seconds']:.1f} s; {s['drawn']} drawn at "
                           f"{1e3 * s['s_per_drawn']:.3f} ms each, skipped at {1e6 * s['s_per_skipped']:.1f} us each; frame = "
                           f"{s['total_drawn']} drawn + rest skipped, extrapolated {s['extrapolated_frame_s']:.0f} s "
                           f"(excludes the vectorised preprocessing)"),
            }

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's untimed extras before the communicator goes away
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
