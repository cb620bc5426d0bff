#!/usr/bin/env python3
"""bench.py — frames/s of the MI355X forward rasterizer on BASELINE.json's metric workload.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

A step = one complete 1920x1080 frame of the synthetic stand-in for MipNeRF-360 'bicycle' — the scene BASELINE.json's
metric is quoted on (mip360_like(6_131_954, seed 361), ring camera 0) — in the reference's own
arithmetic: fp32 coefficients, fp32 frame, reference_compat, every gaussian blended (no approximate early termination):
preprocess -> depth sort -> tile binning -> blend, scene resident in HBM before the timed region — uploaded by the loader along a
Morton curve of the gaussians' means (the loaders' default, --scene-order; the `file_order` leg is the same frame from file-order arrays).  With N > 1 the SAME
frame is sharded by interleaved tile rows over the N GPUs and gathered to rank 0 over RCCL (strong scaling: total work
per frame fixed).  `value` is throughput: frames go through libgsr --views-per-launch at a time (default 4: gsr_render_batch puts
them through ONE preprocess / sort / blend launch sequence — a quarter of the dispatches per frame, each four times better filled,
the scene's geometry read once for the four), and --frames-in-flight such batches (default 2) are in flight per GPU, each on its own
HIP stream with its own workspace (renderer.FramesInFlight); every frame is complete and bit-identical to a single-view,
single-stream render (tests/test_gpu_parity.py); `single_stream` carries the same loop with ONE frame per launch sequence and one in
flight (the per-frame latency), and the per-stage times / roofline are measured that way too.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel (blend): algorithmic bytes 40*E + 12*P + 8*tiles per launch / its
                measured average duration (HIP events on the launch stream) vs 8 TB/s, plus the figures of the pipes
                that actually limit it (VALU issue, LDS) from the committed rocprofv3 counters
  cpu_baseline  the reference's per-gaussian torch loop, ported (oracle/torch_loop.py), timed on this host on a
                bounded subsample of the same frame and extrapolated to the frame
and, as extra keys that are never `value`: PSNR of the timed configuration against the CPU oracle at full size,
per-stage times, counters, and more legs on the same GPU — `configs2` (BASELINE configs[2]: the same scene with fp16 SH storage
and a bf16 frame store), `early_out` (blend stops a wave at T < 1e-4), `camera_set` (a different camera of the 25-pose ring every
frame), `file_order` (the scene's arrays uploaded in file order
instead of along a Morton curve: what rounds 1-2 measured), `garden` (configs[1] stand-in) and `box4k` (configs[4]: 20 M
gaussians at 3840x2160, with its own roofline object).
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
    "bicycle": ("mip360_like", 6_131_954, 361, 1920, 1080, "synthetic stand-in for MipNeRF-360 bicycle (the metric's scene; configs[2]'s storage options are the `configs2` leg)"),
    "garden": ("mip360_like", 5_834_784, 360, 1920, 1080, "synthetic stand-in for MipNeRF-360 garden (configs[1])"),
    "box4k": ("uniform_box", 20_000_000, 20, 3840, 2160, "20M uniform gaussians at 4K (configs[4])"),
}
METRIC = {
    "bicycle": "frames/sec @1080p + PSNR vs torch ref, MipNeRF-360 bicycle, 1/2/4/8 GPUs",
    "garden": "frames/sec @1080p + PSNR vs torch ref, MipNeRF-360 garden (configs[1]; NOT the headline scene), 1/2/4/8 GPUs",
    "box4k": "frames/sec @4K, synthetic 20M gaussians (configs[4])",
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VALU_NS, EXP_NS = 1.1, 3.4  # measured issue cost per wave64 instruction per SIMD, 8 waves per SIMD (tools/valu_microbench.hip)


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
    ap.add_argument("--workload", default="bicycle", choices=sorted(WORKLOADS))
    ap.add_argument("--gaussians", type=int, default=0, help="override the gaussian count (0 = the workload's)")
    ap.add_argument("--early-out-T", type=float, default=0.0)
    ap.add_argument("--colour-stage", type=int, default=0, help="GsrOptions.colour_stage: 0 = sh_to_rgb in the blend's staging (default), 1 = in the preprocess")
    ap.add_argument("--blend-impl", type=int, default=0, help="0 default blend, 1 the same with its walk in plain C")
    ap.add_argument("--sh-half", action="store_true", help="headline with SH coefficients stored as fp16 (default: fp32, the reference's type)")
    ap.add_argument("--bf16-output", action="store_true", help="headline with the frame stored as bfloat16; accumulation stays fp32")
    ap.add_argument("--camera", type=int, default=0)
    ap.add_argument("--camera-set", default="single", choices=["single", "all"],
                    help="single: every step renders --camera; all: steps cycle over the whole camera set (configs[3])")
    ap.add_argument("--input_dir", default="", help="real data: MipNeRF-360 scene dir with sparse/0/{images,cameras}.bin")
    ap.add_argument("--trained_model_path", default="", help="real data: INRIA model dir (point_cloud/iteration_30000/point_cloud.ply)")
    ap.add_argument("--legs", default="configs2,early_out,camera_set,shards8,file_order,garden,box4k", help="extra legs at N=1 (comma list; '' = none)")
    ap.add_argument("--scene-order", default="morton", choices=["morton", "file"],
                    help="how the loader lays the gaussians' arrays out in HBM: along a Morton curve of their means (renderer.GaussianScene "
                         "spatial_order=True: the same frame up to the mutual order of gaussians at exactly equal depth, which the reference "
                         "leaves undefined) or in file order; the `file_order` leg reports the other one")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="independent batches of frames in flight per GPU, each on its own HIP stream and workspace (1 = one stream; the "
                         "single-stream figure is reported beside the headline either way)")
    ap.add_argument("--views-per-launch", type=int, default=None,
                    help="frames per launch sequence of libgsr (gsr_render_batch; 1 = every frame its own ~22 dispatches, rounds 1-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def build_workload(name, args, n_override=0):
    from gsr_amd import synthetic

    gen, n, seed, W, H, desc = WORKLOADS[name]
    if n_override:
        n = n_override
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


def timed_frames(R, cams, opts, out, steps, warmup, dev, slots=1, views=1):
    """Single-GPU legs: `warmup` untimed frames, then `steps` frames between two device synchronisations, `views` frames per launch
    sequence and `slots` such batches in flight like the headline loop (own workspaces and frame buffers of `out`'s type; `R` lends
    its bounds)."""
    from gsr_amd import renderer

    fif = renderer.FramesInFlight(R.scene, slots=slots, max_pairs=R.max_pairs, views=views)
    fif.set_sort_passes(R.sort_passes)
    opts = R.bounded(opts)  # the depth-sort bound the probing frames have learned; fif.stats() below speaks for EVERY frame of a slot
    outs = [torch.empty((views,) + tuple(out.shape), dtype=out.dtype, device=out.device) for _ in range(slots)]
    state = {"f": 0}

    def run(frames):
        done, b = 0, 0
        while done < frames:
            k = min(views, frames - done)
            cs = [cams[(state["f"] + j) % len(cams)] for j in range(k)]
            if views == 1:
                fif.submit(cs[0], opts, out=outs[b % slots][0], slot=b % slots)
            else:
                fif.submit_batch(cs, opts, out=outs[b % slots][:k], slot=b % slots)
            state["f"] += k
            done += k
            b += 1

    run(slots * views)  # setup: every slot's stream and workspace used once (see main) before the `warmup` frames
    run(warmup)
    state["f"] = 0
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    for k in range(slots):
        fif.stats(k)  # raises if ANY frame of the slot exceeded max_pairs or the depth-sort bound (GsrOptions.keep_flags chain)
    return el


def psnr_pair(img, ref):
    mse = float(np.mean((img.astype(np.float64) - ref) ** 2))
    return (None if mse == 0 else 10 * np.log10(1.0 / mse)), float(np.abs(img - ref).max())


def pmc_profile(workload):
    """Per-launch rocprofv3 counters of the blend kernel, committed under profiles/ (tools/summarize_profiles.py)."""
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(tfile)).get(workload, {})
    except Exception:
        return {}


def stage_profile(R, scene, cam, opts, out_shape, tiles, reps, sh_half, prof, dev):
    """Per-stage times of one frame (HIP events around gsr_preprocess / gsr_bin_sort / gsr_blend on ONE stream, mean over `reps`
    frames after 3 untimed ones) and the roofline object of the dominant kernel, the blend.

    roofline.achieved / peak / frac are the CONTRACT figure: algorithmic bytes 40 E + 12 P + 8 tiles (+ 216 per colour
    the blend evaluated itself: the SH row SURVEY prices at 192 B, the mean, the write-back) per launch
    over the kernel's time against 8 TB/s.  The kernel is not bound by HBM but by vector-ALU issue (exact per-pixel evaluation:
    ~13 issue slots per evaluated (8x8 quadrant, entry), a quarter-rate v_exp_f32 among them), so `bound` says "valu" and the
    fractions that describe it ride along: valu_issue_frac (VALU wave-instructions per launch, from the committed rocprofv3
    counters, at one issue per 2 cycles per SIMD at the profiled clock) and valu_frac_calibrated (the same instructions priced
    with tools/valu_microbench.hip's measured rates on this chip: 1.1 ns per plain VALU wave-instruction per SIMD, 3.4 ns per
    v_exp_f32 — what a kernel made of nothing else would need)."""
    import ctypes as C

    from gsr_amd._lib import check, lib

    n = scene.n
    W, H = cam.width, cam.height
    ws = R._workspace(W, H)
    sc = scene.c_struct()
    out = torch.empty(out_shape, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = int(stream.cuda_stream)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for i in range(reps + 3):
        e = ev[max(i - 3, 0)]
        e[0].record(stream)
        check(lib.gsr_preprocess(C.byref(sc), C.byref(cam), C.byref(opts), ws.data_ptr(), ws.numel(), None, sp))
        e[1].record(stream)
        check(lib.gsr_bin_sort(n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), sp))
        e[2].record(stream)
        check(lib.gsr_blend(C.byref(sc), n, C.byref(cam), C.byref(opts), R.max_pairs, ws.data_ptr(), ws.numel(), out.data_ptr(), None, sp))
        e[3].record(stream)
    torch.cuda.synchronize(dev)
    st = R.stats()
    stage = [float(np.mean([e[k].elapsed_time(e[k + 1]) for e in ev])) for k in range(3)]
    E, P, V = st["fetched_entries"], out.shape[0] * out.shape[1], st["n_visible"]  # E = entries actually fetched
    # + the colour of every gaussian the blend evaluated itself (GsrOptions.colour_stage = 0: the 192-B SH row — 96 B as fp16 — and
    # the 12-B mean are read when a tile first stages the gaussian, 12 B of colour written back): those bytes left the preprocess
    C_ev = st.get("colour_evals", 0)
    blend_bytes = 40.0 * E + 12.0 * P + 8.0 * tiles + ((96.0 if sh_half else 192.0) + 12.0 + 12.0) * C_ev
    # preprocess: 44 B of geometry per gaussian; the outputs (48-B record + key + packed rect + the slack of the 64 B SURVEY.md §8(d)
    # allows) only for the V visible ones; the SH row (192 B, 96 B as fp16) only where the preprocess evaluates colours
    # (colour_stage = 1) and then only for visible gaussians (rounds 1-3 priced every gaussian at 236 B: a figure that could
    # exceed the HBM peak)
    pre_bytes = 44.0 * n + 64.0 * V + (0.0 if C_ev else (96.0 if sh_half else 192.0) * V)
    # bin + sort, as THIS design moves them: depth sort of the V visible keys (pass 0 reads all N keys twice —
    # histogram, scatter — and writes V (key, id, packed rect) triples; each later pass reads V keys for its histogram, then reads
    # and writes the V triples), pair count / emit (reads the V ids + rects, writes D (cell key, value) pairs), the 2-pass cell
    # sort (per pass: D keys for the histogram, D pairs read and written), the range scan over the D sorted keys
    D, passes = st["n_pairs_bbox"], max(1, st["sort_passes"])
    sort_bytes = 8.0 * n + 12.0 * V + (passes - 1) * 28.0 * V + 12.0 * V + 8.0 * D + 2 * 20.0 * D + 4.0 * D + 8.0 * tiles
    # SURVEY.md §8(d)'s own formula for the stage with its ideal single pass: 8 N (scan) + 12 D (emit key8 + val4) + 2 * 12 * D * passes
    # (passes = 1) + 8 D (range scan) + 8 tiles — the lower bound the stage is held against beside the bytes this design moves
    sort_ideal = 8.0 * n + 12.0 * D + 2 * 12.0 * D * 1 + 8.0 * D + 8.0 * tiles
    achieved = blend_bytes / (stage[2] * 1e-3) / 1e9
    # SURVEY.md §8(d)'s blend line exactly as written — 40 E + 12 P + 8 tiles — beside the figure above, which also books the bytes of
    # the deferred colours (216 per colour evaluated in this launch: work §8(d) prices in the PREPROCESS line and this design moved here)
    survey_bytes = 40.0 * E + 12.0 * P + 8.0 * tiles
    roof = {
        "kernel": "gsr::blend_walk_kernel", "bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": prof.get("blend_kernel_bytes_per_launch"),
        "algorithmic_bytes_per_launch": blend_bytes, "avg_kernel_ms": stage[2],
        "frac_survey_8d": survey_bytes / (stage[2] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_survey_8d": survey_bytes,
        "bytes_formula": "frac: 40 E_staged + 12 P + 8 tiles + 216 colour_evals (192-B SH row + 12-B mean + 12 B written back per colour the "
                         "blend evaluated itself; 120 with fp16 SH); frac_survey_8d: 40 E_staged + 12 P + 8 tiles only",
        "limiter": "vector-ALU issue (exact per-pixel evaluation: ~70 flop and 6.4 exp per algorithmic byte, SURVEY.md §7 hard part 1): "
                   "achieved / peak / frac are the HBM contract figure, valu_* the fractions of the pipe that bounds the kernel",
        "note": "avg_kernel_ms: HIP events around the stage on ONE stream, one frame in flight (profiles/*kernel_stats.csv, collected "
                "with --frames-in-flight 1, agrees); in the timed loop kernels of the frames in flight share the machine and "
                "their individual durations stretch.  traffic = rocprofv3 FETCH_SIZE x2 (gfx950 correction, an upper bound for "
                "gathered 48-B records) + WRITE_SIZE, separate --pmc passes, from profiles/",
    }
    # honesty figures: pixel evaluations, and the pipes that bound the kernel, priced from the committed counters
    roof["pixel_evaluations"] = 64.0 * st["wave_entries"]
    if prof.get("valu_insts_per_wave_entry"):
        clk = prof.get("clock_ghz", 2.0)
        valu = st["wave_entries"] * prof["valu_insts_per_wave_entry"]  # wave-instructions per launch (scales with the evaluated entries)
        t = stage[2] * 1e-3
        roof["valu_issue_frac"] = valu * 2.0 / t / (1024 * clk * 1e9)
        # calibrated: every VALU wave-instruction at the microbenchmark's 1.1 ns per SIMD, + 2.3 ns more for each evaluation's v_exp_f32
        roof["valu_frac_calibrated"] = (valu * VALU_NS + st["wave_entries"] * (EXP_NS - VALU_NS)) * 1e-9 / 1024 / t
        roof["valu_calibration"] = {"plain_valu_ns_per_wave_instr_per_simd": VALU_NS, "v_exp_f32_ns": EXP_NS,
                                    "source": "tools/valu_microbench.hip on MI355X, 8 waves per SIMD (profiles/r3_valu_microbench.txt)"}
    if prof.get("lds_busy_frac") is not None:
        roof["lds_busy_frac_profiled"] = prof["lds_busy_frac"]
    copy_gbs = measured_copy_gbs(dev)

    def stage_roof(kernel, bound, nbytes, ms, traffic, note):
        a = nbytes / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": bound, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": nbytes, "avg_kernel_ms": ms,
                "frac_of_measured_copy": a / copy_gbs if copy_gbs else None, "note": note}

    stage_roofs = {
        "preprocess": stage_roof("gsr::preprocess_kernel", "hbm", pre_bytes, stage[0], prof.get("preprocess_bytes_per_launch"),
                                 "44 N + 64 V bytes (+ 192 V when the preprocess evaluates the colours, GsrOptions.colour_stage = 1): geometry of "
                                 "every gaussian, the outputs of the visible ones"),
        "bin_sort": stage_roof("gsr::radix_* + pair_* + tile_* (the stage's ~20 dispatches together)", "hbm", sort_bytes, stage[1],
                               prof.get("bin_sort_bytes_per_frame"),
                               "bytes as this design moves them (bench.py stage_profile); frac_survey_8d / ideal_bytes_survey_8d: SURVEY.md "
                               "§8(d)'s formula with passes = 1, the stage's lower bound"),
    }
    stage_roofs["bin_sort"]["ideal_bytes_survey_8d"] = sort_ideal
    stage_roofs["bin_sort"]["frac_survey_8d"] = sort_ideal / (stage[1] * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["hbm_copy_gbs_measured"] = copy_gbs
    return {"roofline": roof, "stage_rooflines": stage_roofs, "hbm_copy_gbs_measured": copy_gbs,
            "stage_ms": {"preprocess": stage[0], "bin_sort": stage[1], "blend": stage[2]}, "stats": st}


_COPY_GBS = {}


def measured_copy_gbs(dev):
    """What a plain device-to-device copy of 1 GiB sustains on THIS GPU in THIS run (read + write bytes over the time of the best of
    5): the practical ceiling of an HBM-bound kernel, printed beside the 8 TB/s spec the contract fractions are taken against."""
    key = str(dev)
    if key not in _COPY_GBS:
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        best = float("inf")
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        _COPY_GBS[key] = 2.0 * a.numel() * 4 / (best * 1e-3) / 1e9
        del a, b
    return _COPY_GBS[key]


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

    real = bool(args.input_dir and args.trained_model_path)
    if real:
        cols, cam_list, n, W, H, desc = build_real_workload(args)
    else:
        cols, cam_list, n, W, H, desc = build_workload(args.workload, args, args.gaussians)
    cam_args = cam_list[0]
    packed = utils.pack_gaussians(cols)
    del cols
    spatial = args.scene_order == "morton"
    scene = renderer.GaussianScene.from_packed(packed, device=dev, sh_half=args.sh_half, spatial_order=spatial)
    cams = [renderer.make_camera(*c) for c in cam_list]
    cam = cams[0]
    ncam = len(cams)
    order_warm_ms = None
    if spatial and scene.n:  # what the ordering costs once everything it uses is loaded: the permutation + one gather of the arrays
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        perm = renderer.scene_order(scene.t["means"])
        tmp = [scene.t[k].index_select(0, perm) for k in scene.FIELDS]
        e1.record()
        e1.synchronize()
        order_warm_ms = float(e0.elapsed_time(e1))
        del perm, tmp
    plan = gdist.TileRowPlan(H, W, world)
    out_dtype = torch.bfloat16 if args.bf16_output else torch.float32
    # Views per launch sequence K and batches in flight S (tools/batch_timing.py, bench frame, a different camera every frame /
    # one camera: K x S = 1 x 3: 1789 / 1858 frames/s (rounds 1-4: one view per launch sequence), 2 x 2: 1906 / 1965, 4 x 2: 2016 / 2089,
    # 8 x 3: 2047 / 2089; this command line with --steps 20 / 100, two runs each (gpurun_out of round 5, frames/s): 4 x 2: 2023-2075 /
    # 2120-2174, 5 x 2: 2099-2115 / 2182-2184, 6 x 3: 2129-2133 / 2204-2237, 7 x 3: 2102-2177 / 2146-2235, 8 x 3: 2074-2167 / 2137-2219):
    # 6 x 3 (18 slices of 0.69 GB).  Tile-row shards of 8: 4 x 3.
    K = max(1, args.views_per_launch if args.views_per_launch is not None else (6 if world == 1 else 4))
    S = max(1, args.frames_in_flight if args.frames_in_flight is not None else 3)
    fif = renderer.FramesInFlight(scene, slots=S, views=K)
    R = fif.rasterizers[0]
    state = {"f": 0, "b": 0}
    if world == 1:  # no sharding: blend straight into a frame buffer per slot
        opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, output_bf16=args.bf16_output, colour_stage=args.colour_stage)
        frames = [torch.zeros((K, H, W, 3), dtype=out_dtype, device=dev) for _ in range(S)]
        strip_view = frames[0][0]

        def step(count):
            """Enqueue the next `count` <= K frames as one batch."""
            f, b = state["f"], state["b"]
            state["f"] += count
            state["b"] += 1
            cs = [cams[(f + j) % ncam] for j in range(count)]
            if K == 1:
                fif.submit(cs[0], opts, out=frames[b % S][0], slot=b % S)
            else:
                fif.submit_batch(cs, opts, out=frames[b % S][:count], slot=b % S)
            return frames[b % S][count - 1]

        def drain():
            return None
    else:
        opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, output_bf16=args.bf16_output,
                                     colour_stage=args.colour_stage, **plan.shard_options(rank))
        # gdist.ShardedFrames: batch b renders on stream b % S into wire buffer b % S, its strips are gathered asynchronously
        # over RCCL (one collective per batch) while the next batches render, frames are finished in order on the main stream

        def render_strips(k, c, strips):
            if K == 1:
                fif.rasterizers[k].enqueue(c, opts, out=strips)
            else:
                fif.rasterizers[k].enqueue_batch(c, opts, out=strips[: len(c)])

        sf = gdist.ShardedFrames(plan, rank, dev, S, render_strips, dtype=out_dtype, streams=fif.streams, views=K)
        strip_view = sf.fg.own_view(0) if K == 1 else sf.fg.own_view(0)[0]

        def step(count):
            f = state["f"]
            state["f"] += count
            state["b"] += 1
            cs = [cams[(f + j) % ncam] for j in range(count)]
            return sf.submit(cs[0] if K == 1 else cs)

        def drain():
            return sf.drain()

    def run(step_fn, frames_to_go, k):
        last = None
        while frames_to_go > 0:
            c = min(k, frames_to_go)
            last = step_fn(c)
            frames_to_go -= c
        return last

    # size the pair buffer to the heaviest view once (grows on overflow), outside the timed region
    need = max(R.fit_pairs(c, opts) for c in cams)
    fif.set_max_pairs(need)
    fif.set_sort_passes(R.sort_passes)  # the depth-sort bound learned from the views' counters (GsrOptions.depth_sort_passes)
    opts = R.bounded(opts)  # passed explicitly: enqueue() / submit() apply no bound of their own; every slot's stats() after the timed
    # region speaks for ALL its frames (they are chained with GsrOptions.keep_flags)
    R.render(cam, opts, out=strip_view)
    shard_stats = dict(R.last_stats)
    torch.cuda.synchronize(dev)
    # setup, like the probing frames above: one untimed batch through EVERY slot, so that each slot's stream (its hardware queue is
    # created at first use) and workspace have been used before the W warmup steps, whatever W is
    run(step, S * K, K)
    drain()
    state["f"] = state["b"] = 0
    torch.cuda.synchronize(dev)

    def timed_region(step_fn, drain_fn, steps, warmup, k, st):
        run(step_fn, warmup, k)
        drain_fn()
        st["f"] = st["b"] = 0
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        last = run(step_fn, steps, k)  # EXACTLY `steps` frames: batches of k, the last one may hold fewer
        if world > 1:
            last = drain_fn()  # the last frames' gathers complete inside the timed region
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, last

    elapsed, frame = timed_region(step, drain, args.steps, args.warmup, K, state)
    for k in range(S):
        if fif.rasterizers[k]._ws is not None:
            fif.stats(k)  # raises if ANY of the slot's frames exceeded max_pairs or the depth-sort bound
    # the same loop with ONE frame per launch sequence and ONE in flight (one stream, one workspace): the per-frame latency figure
    single = None
    if S > 1 or K > 1:
        one = renderer.FramesInFlight(scene, slots=1, max_pairs=need)
        one.set_sort_passes(R.sort_passes)
        st1 = {"f": 0, "b": 0}
        sf1 = None
        one_frame = torch.zeros((H, W, 3), dtype=out_dtype, device=dev) if world == 1 else None
        if world > 1:
            sf1 = gdist.ShardedFrames(plan, rank, dev, 1, lambda k, c, strip: one.rasterizers[0].enqueue(c, opts, out=strip),
                                      dtype=out_dtype, streams=one.streams)

        def step1(count):
            f = st1["f"]
            st1["f"] += 1
            if world == 1:
                one.submit(cams[f % ncam], opts, out=one_frame, slot=0)
                return None
            return sf1.submit(cams[f % ncam])

        def drain1():
            return sf1.drain() if sf1 is not None else None

        el1, _ = timed_region(step1, drain1, args.steps, args.warmup, 1, st1)
        single = {"frames_per_s": args.steps / el1, "ms_per_frame": 1e3 * el1 / args.steps,
                  "note": "the same timed loop with one frame per launch sequence and one in flight (one stream, one workspace): per-frame latency"}
        del one, sf1, one_frame

    result = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        result = {
            "metric": METRIC[args.workload] if not real else METRIC["bicycle"],
            "value": args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",  # the arithmetic type of the whole path; storage options are named in config
            "data": "real" if real else "synthetic",
            "config": {"workload": f"{'real' if real else args.workload}: {desc}", "gaussians": n, "width": W, "height": H,
                       "camera": args.camera if ncam == 1 else f"cycling over {ncam} cameras",
                       "sharding": f"tile rows interleaved over {world} GPU(s), RCCL gather to rank 0" if world > 1 else "none",
                       "scene_order": ("morton curve of the means: the loaders' default (GaussianScene.sort_spatially, on the GPU at upload); "
                                       "same frame up to exact depth ties") if spatial else "file (spatial_order=False)",
                       "scene_order_ms_at_upload": scene.order_ms,
                       "scene_order_ms_warm": order_warm_ms,
                       "scene_order_note": "at_upload: events around gsr_scene_order + the gather of the five arrays the FIRST time in this "
                                           "process — it includes loading libgsr's code objects and torch's gather kernels and the caching "
                                           "allocator's first 2 GB; warm: the same two steps repeated on the resident scene",
                       "reference_compat": True, "early_out_T": args.early_out_T, "depth_sort_passes": R.sort_passes, "sh_storage": "f16" if args.sh_half else "f32",
                       "frame_storage": "bf16 (fp32 accumulation)" if args.bf16_output else "f32",
                       "blend_impl": {0: "valu", 1: "valu, plain-C walk"}.get(args.blend_impl, str(args.blend_impl))},
            "stats_rank0_shard": shard_stats,
        }
        result["config"]["views_per_launch"] = K
        result["config"]["frames_in_flight"] = S
        result["config"]["throughput_mode"] = (f"{K} frames per launch sequence (gsr_render_batch), {S} such batches in flight on separate streams; "
                                               "every frame bit-identical to a single-view render")
        if single is not None:
            result["single_stream"] = single

    # ---- rank 0: per-stage timing + roofline of ITS shard (the whole frame at N=1), outside the timed region; at N=1 also
    # ---- PSNR vs the oracle, the extra legs and the CPU baseline -------------------------------------------------------
    if rank == 0:
        if world == 1:
            frame = R.enqueue(cam, opts, out=strip_view)  # the frame checked against the oracle below is camera 0's (single-view path)
        full_opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, depth_sort_passes=R.sort_passes, colour_stage=args.colour_stage,
                                          **(plan.shard_options(rank) if world > 1 else {}))
        prof = stage_profile(R, scene, cam, full_opts, plan.strip_shape(rank) if world > 1 else (H, W, 3),
                             len(plan.rows[rank]) * ((W + 15) // 16) if world > 1 else ((W + 15) // 16) * ((H + 15) // 16),
                             max(10, min(50, args.steps)), args.sh_half, pmc_profile(args.workload) if world == 1 else {}, dev)
        result.update(prof)
        if world > 1:
            result["roofline"]["note"] = "rank 0's shard (interleaved tile rows); " + result["roofline"]["note"]

    if rank == 0 and world == 1:
        legs = [x for x in args.legs.split(",") if x] if not real else []
        leg_imgs = {}
        steps_leg, warm_leg = args.steps, min(args.warmup, 5) + 1
        leg_out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)

        # (1) the same frame with the usual 3DGS saturation cut-off (INRIA's T < 1e-4): the headline stays the exact mode (no
        # blend work skipped, reference semantics Q5); this shows what north_star's "ballot early-out on saturated alpha" buys
        # inside its PSNR >= 50 dB tolerance
        if "early_out" in legs and args.early_out_T == 0.0:
            eo_opts = renderer.make_options(early_out_T=1e-4, blend_impl=args.blend_impl)
            el = timed_frames(R, cams, eo_opts, leg_out, steps_leg, warm_leg, dev, S, K)
            R.enqueue(cam, eo_opts, out=leg_out)
            s = R.stats()
            leg_imgs["early_out"] = leg_out.cpu().numpy()
            result["early_out"] = {"early_out_T": 1e-4, "frames_per_s": steps_leg / el, "ms_per_step": 1e3 * el / steps_leg,
                                   "wave_entries": s["wave_entries"], "fetched_entries": s["fetched_entries"],
                                   "note": "not the headline: same workload with the blend stopping a wave once T < 1e-4 for its 64 pixels"}

        # (1b) the same scene over the whole 25-camera ring, a different view every frame (configs[3]'s camera set on one GPU): the
        # launch-order hint of a slot then comes from a view six cameras away, the depth-sort bound and pair buffer from the heaviest view
        if "camera_set" in legs and ncam == 1 and not real:
            from gsr_amd import synthetic

            fx_ = synthetic.pinhole_focal(W)
            ring = [renderer.make_camera(p.qvec, p.tvec, 2 * fx_, 2 * fx_, 2 * W, 2 * H, W, H) for p in synthetic.ring_cameras(25)]
            Rc = renderer.Rasterizer(scene, max_pairs=R.max_pairs)
            c_opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, colour_stage=args.colour_stage)
            Rc.max_pairs = max(Rc.fit_pairs(c, c_opts) for c in ring)
            el = timed_frames(Rc, ring, c_opts, leg_out, max(steps_leg, 50), warm_leg, dev, S, K)
            result["camera_set"] = {"cameras": len(ring), "frames_per_s": max(steps_leg, 50) / el, "ms_per_step": 1e3 * el / max(steps_leg, 50),
                                    "note": "not the headline: the same scene, a different camera of the 25-pose ring every frame (one GPU)"}
            del Rc

        # (1c) one-GPU rehearsal of configs[3]'s 8-GPU sharding: the slowest and a typical rank's tile-row shard of the headline frame
        # (dist.TileRowPlan's default plan; the loop a rank of `bench.py --gpus 8` runs — four views per launch sequence, three batches
        # in flight — without the gather): what bounds the frame rate of the 8-GPU run from the render side
        if "shards8" in legs and not real:
            plan8 = gdist.TileRowPlan(H, W, 8)
            heavy = max(range(8), key=lambda r: (len(plan8.rows[r]), r))
            per_rank = {}
            for r in sorted({0, heavy}):
                s_opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl, colour_stage=args.colour_stage,
                                               **plan8.shard_options(r))
                Rs = renderer.Rasterizer(scene)
                Rs.max_pairs = max(Rs.fit_pairs(c, s_opts) for c in cams)
                s_steps = max(steps_leg, 120)  # thirty batches: the fill and drain of the three slots weigh a few per cent
                el = timed_frames(Rs, cams, s_opts, torch.empty(plan8.strip_shape(r), dtype=torch.float32, device=dev), s_steps, warm_leg, dev, 3, 4)
                per_rank[r] = 1e3 * el / s_steps
                del Rs
            worst = max(per_rank.values())
            result["shards8"] = {"ms_per_frame_by_rank": {str(r): v for r, v in per_rank.items()}, "tile_rows_by_rank": {str(r): len(plan8.rows[r]) for r in per_rank},
                                 "slowest_ms_per_frame": worst, "frames_per_s_bound_at_8_gpus": 1e3 / worst, "views_per_launch": 4, "frames_in_flight": 3,
                                 "note": "not the headline and not a multi-GPU measurement: ONE GPU renders rank 0's (the gather's root, eight tile rows) and the "
                                         "heaviest rank's (ten tile rows) shard of the headline frame the way a rank of `bench.py --gpus 8` does, gather "
                                         "excluded — the render-side bound of the 8-GPU frame rate (DESIGN.md section 6)"}

        # (2) BASELINE configs[2]: fp16 SH storage + bf16 frame store (accumulation stays fp32: bf16 accumulators measure 41 dB,
        # below the 50 dB bar).  PSNR below is against the fp32 oracle of the fp32 coefficients.
        if "configs2" in legs and not args.sh_half:
            scene_h = renderer.GaussianScene.from_packed(packed, device=dev, sh_half=True, spatial_order=spatial)
            Rh = renderer.Rasterizer(scene_h, max_pairs=R.max_pairs)
            h_opts = renderer.make_options(output_bf16=True)
            h_out = torch.empty((H, W, 3), dtype=torch.bfloat16, device=dev)
            el = timed_frames(Rh, cams, h_opts, h_out, steps_leg, warm_leg, dev, S, K)
            Rh.enqueue(cam, h_opts, out=h_out)
            s = Rh.stats()
            leg_imgs["configs2"] = h_out.float().cpu().numpy()
            result["configs2"] = {"frames_per_s": steps_leg / el, "ms_per_step": 1e3 * el / steps_leg, "sh_storage": "f16",
                                  "frame_storage": "bf16 (fp32 accumulation)", "fetched_entries": s["fetched_entries"],
                                  "note": "not the headline: BASELINE configs[2] storage options on the same scene and camera"}
            del Rh, scene_h, h_out

        # (2b) the same scene uploaded in the OTHER order (headline: along a Morton curve of the means — waves are culled whole, the SH
        # rows of the visible gaussians are contiguous, the blend's record gathers hit L2 more often; this leg: file order, as the
        # reference reads the .ply and as rounds 1-2 measured).  The frame is the same up to the mutual order of gaussians at exactly
        # equal depth, which the reference leaves undefined.
        if "file_order" in legs and not args.sh_half:
            scene_m = renderer.GaussianScene.from_packed(packed, device=dev, spatial_order=not spatial)
            Rm = renderer.Rasterizer(scene_m, max_pairs=R.max_pairs)
            m_opts = renderer.make_options(early_out_T=args.early_out_T, blend_impl=args.blend_impl)
            m_out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            Rm.render(cam, m_opts, out=m_out)  # learns the depth-sort bound
            el = timed_frames(Rm, cams, m_opts, m_out, steps_leg, warm_leg, dev, S, K)
            mprof = stage_profile(Rm, scene_m, cam, Rm.bounded(m_opts), (H, W, 3), ((W + 15) // 16) * ((H + 15) // 16), 10, False, {}, dev)
            Rm.enqueue(cam, m_opts, out=m_out)
            leg_imgs["file_order"] = m_out.cpu().numpy()
            result["file_order"] = {"scene_order": "file" if spatial else "morton", "frames_per_s": steps_leg / el, "ms_per_step": 1e3 * el / steps_leg,
                                    "stage_ms": mprof["stage_ms"], "stats": mprof["stats"],
                                    "note": "not the headline: the same scene, camera and arithmetic with the gaussians' arrays uploaded in the other "
                                            "order (file order when the headline is Morton order, the loader's spatial_order option, and vice versa)"}
            del Rm, scene_m, m_out

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
                result["psnr_vs_oracle_db"], result["max_abs_vs_oracle"] = psnr_pair(frame.float().cpu().numpy(), oracle_img)
                for k, im in leg_imgs.items():
                    result[k]["psnr_vs_oracle_db"], result[k]["max_abs_vs_oracle"] = psnr_pair(im, oracle_img)
                result["cpu_oracle_c"] = {"frame_s": t_pre + t_comp, "threads": threads, "drawn": int(drawn),
                                          "note": "oracle/gsr_oracle.c, OpenMP over x bands; checker, not the baseline"}
        if not args.no_cpu_baseline:
            from oracle import torch_loop

            cores = host_threads()
            torch.set_num_threads(cores)  # the reference sets cpu_count()-1 (rasterize.py:323)
            s = torch_loop.timed_sample(pre, order, W, H, budget_s=args.cpu_budget_s, max_gaussians=400_000)
            result["cpu_baseline"] = {
                "value": 1.0 / s["extrapolated_frame_s"], "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": (f"reference per-gaussian torch loop (oracle/torch_loop.py, checked against the reference's frames in "
                           f"tests/test_oracle_golden.py): uniform random sample of {s['sampled']} of the "
                           f"frame's {s['total_iterations']} loop iterations, {s['seconds']:.1f} s; {s['drawn']} drawn at "
                           f"{1e3 * s['s_per_drawn']:.3f} ms each, skipped at {1e6 * s['s_per_skipped']:.1f} us each; frame = "
                           f"{s['total_drawn']} drawn + rest skipped, extrapolated {s['extrapolated_frame_s']:.0f} s "
                           f"(excludes the vectorised preprocessing)"),
            }
            result["vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]  # not vs_baseline: no published number exists
        del pre, order, oracle_img, leg_imgs

        # (3) BASELINE configs[1]: the garden stand-in, fp32, exact — last, once the headline scene's host arrays are gone
        if "garden" in legs and args.workload != "garden":
            del R, scene, packed
            torch.cuda.empty_cache()
            gcols, gcam_list, gn, _, _, gdesc = build_workload("garden", args)
            gscene = renderer.GaussianScene.from_columns(gcols, device=dev, spatial_order=spatial)
            del gcols
            gcams = [renderer.make_camera(*c) for c in gcam_list]
            Rg = renderer.Rasterizer(gscene)
            g_opts = renderer.make_options()
            Rg.max_pairs = max(Rg.fit_pairs(c, g_opts) for c in gcams)
            el = timed_frames(Rg, gcams, g_opts, leg_out, steps_leg, warm_leg, dev, S, K)
            Rg.enqueue(gcams[0], g_opts, out=leg_out)
            s = Rg.stats()
            result["garden"] = {"frames_per_s": steps_leg / el, "ms_per_step": 1e3 * el / steps_leg, "gaussians": gn,
                                "stats": s, "note": "not the headline: BASELINE configs[1] stand-in (" + gdesc + "), fp32, exact; "
                                "its oracle parity is tests/test_gpu_configs.py"}

        # (4) BASELINE configs[4] on this GPU: 20 M uniform gaussians at 3840x2160, fp32, exact — its own stage times and its own
        # roofline object (the config asks for "rocprof GB/s vs roofline": profiles/r3_box4k_* hold the rocprofv3 side of it)
        if "box4k" in legs and args.workload != "box4k" and not args.gaussians:
            R = scene = packed = fif = frames = leg_out = strip_view = frame = Rg = gscene = None  # the other scenes leave HBM
            torch.cuda.empty_cache()
            bcols, bcam_list, bn, bW, bH, bdesc = build_workload("box4k", args)
            bscene = renderer.GaussianScene.from_columns(bcols, device=dev, spatial_order=spatial)
            del bcols
            bcam = renderer.make_camera(*bcam_list[0])
            Rb = renderer.Rasterizer(bscene)
            b_opts = renderer.make_options()
            Rb.fit_pairs(bcam, b_opts)
            b_out = torch.empty((bH, bW, 3), dtype=torch.float32, device=dev)
            steps_b = max(5, min(args.steps, 30))
            el = timed_frames(Rb, [bcam], b_opts, b_out, steps_b, 3, dev, min(S, 2), K)
            bprof = stage_profile(Rb, bscene, bcam, Rb.bounded(b_opts), (bH, bW, 3), ((bW + 15) // 16) * ((bH + 15) // 16),
                                  10, False, pmc_profile("box4k"), dev)
            result["box4k"] = {"frames_per_s": steps_b / el, "ms_per_step": 1e3 * el / steps_b, "gaussians": bn, "width": bW, "height": bH,
                               "frames_in_flight": min(S, 2), "views_per_launch": K, **bprof,
                               "note": "not the headline: BASELINE configs[4] (" + bdesc + ") on ONE GPU, fp32, exact; its oracle parity "
                                       "(125.6 dB at full size) is tests/test_gpu_configs.py"}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's untimed extras before the communicator goes away
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
