"""GPU: BASELINE.json's configurations at their FULL sizes, through the C ABI, against the CPU oracle.

  configs[1]  garden stand-in    5 834 784 gaussians, 1920x1080, fp32            frame >= 100 dB; integer rects counted vs the oracle
  configs[2]  bicycle stand-in   6 131 954 gaussians, 1920x1080                  fp32 frame >= 100 dB (the bench headline);
                                                                                 fp16-SH storage vs the oracle fed the fp16-rounded
                                                                                 coefficients >= 100 dB; + bf16 frame store >= 55 dB
  configs[4]  uniform_box        20 000 000 gaussians, 3840x2160                 frame >= 100 dB; footprint culling off and 8-shard
                                                                                 reassembly bit-identical at that size
  configs[3]  (8-GPU tile-row shard + RCCL gather of configs[2]): tools/dist_check.py under torchrun, nccl when the box has
              >= 2 GPUs, otherwise the same ranks over gloo sharing the one GPU (the sharding / gather / assembly logic is the same).

The real MipNeRF-360 scenes are not available offline: these are the seeded stand-ins of SURVEY.md §8(d).
Tolerance stated by north_star: PSNR >= 50 dB vs the torch reference; held here: >= 100 dB (fp32), >= 55 dB (bf16 store).
The oracle takes ~10 s (1080p) / ~40 s (4K) on the box's host cores.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_frames_close, psnr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gsr_amd  # noqa: F401
    from gsr_amd import renderer, synthetic, utils
    from oracle import cpu_oracle as orc

    class NS:
        pass

    ns = NS()
    ns.renderer, ns.synthetic, ns.utils, ns.orc = renderer, synthetic, utils, orc
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return ns


def _ring_camera(G, W, H, pose=0):
    p = G.synthetic.ring_cameras(25)[pose]
    fx = G.synthetic.pinhole_focal(W)
    args = (p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
    return G.renderer.make_camera(*args), G.orc.camera(*args)


def _oracle_frame(G, packed, ocam):
    pre = G.orc.preprocess(packed, ocam)
    order = G.orc.depth_order(pre["cam_means"])
    screen, _, drawn = G.orc.composite(order, pre, ocam.width, ocam.height, threads=G.orc.max_threads())
    return screen.transpose(1, 0, 2), pre, drawn


def test_configs2_bicycle_full_size(G):
    """The bench headline (fp32) and configs[2]'s storage options on the bicycle stand-in at 1080p."""
    W, H = 1920, 1080
    packed = G.utils.pack_gaussians(G.synthetic.mip360_like(6_131_954, 361))
    cam, ocam = _ring_camera(G, W, H)
    mk = G.renderer.make_options

    oimg, _, drawn = _oracle_frame(G, packed, ocam)
    R = G.renderer.Rasterizer(G.renderer.GaussianScene.from_packed(packed))
    img = R.render(cam).cpu().numpy()
    st = dict(R.last_stats)
    assert 0 < st["n_visible"] <= drawn and st["overflow"] == 0
    assert_frames_close(img, oimg)
    print(f"\nbicycle fp32: {psnr(img, oimg):.1f} dB vs oracle, stats {st}")
    del R

    # fp16 SH storage == the fp16-rounded coefficients evaluated in fp32: compare with the oracle fed exactly those
    rounded = dict(packed)
    rounded["sh"] = packed["sh"].astype(np.float16).astype(np.float32)
    oimg16, _, _ = _oracle_frame(G, rounded, ocam)
    Rh = G.renderer.Rasterizer(G.renderer.GaussianScene.from_packed(packed, sh_half=True))
    half = Rh.render(cam).cpu().numpy()
    assert_frames_close(half, oimg16)
    db_vs_fp32 = psnr(half, oimg)
    assert db_vs_fp32 >= 75.0, db_vs_fp32                        # fp16 storage alone vs the fp32-coefficient oracle

    # + bf16 frame store (accumulation fp32): 8 mantissa bits of storage; the bar is 50 dB
    both = Rh.render(cam, mk(output_bf16=True))
    assert both.dtype == torch.bfloat16
    assert torch.equal(both, torch.from_numpy(half).to(torch.bfloat16).to(both.device))   # store = RNE of the fp32 frame
    db = psnr(both.float().cpu().numpy(), oimg)
    assert db >= 55.0, db
    print(f"bicycle fp16 SH: {psnr(half, oimg16):.1f} dB vs oracle(fp16-rounded), {db_vs_fp32:.1f} dB vs oracle(fp32); + bf16 store {db:.1f} dB")


def test_configs1_garden_full_size_and_rect_mismatch_count(G):
    """configs[1] at full size, plus the honest measure of 'integer outputs bit-exact': ocml's expf vs libm's under ceil/floor
    moves a rect by one tile for a gaussian in ~1e5.  Counted here at bench scale, bounded, printed."""
    W, H = 1920, 1080
    N = 5_834_784
    packed = G.utils.pack_gaussians(G.synthetic.mip360_like(N, 360))
    cam, ocam = _ring_camera(G, W, H)
    oimg, pre, drawn = _oracle_frame(G, packed, ocam)
    R = G.renderer.Rasterizer(G.renderer.GaussianScene.from_packed(packed))
    img = R.render(cam).cpu().numpy()
    assert R.last_stats["n_visible"] <= drawn
    assert_frames_close(img, oimg)
    dbg = R.preprocess_debug(cam)
    tb, pb = dbg["tile_bboxes"].cpu().numpy(), dbg["pixel_bboxes"].cpu().numpy()
    del dbg
    for name, got, ref, step in (("tile_bboxes", tb, pre["tile_bboxes"], 1), ("pixel_bboxes", pb, pre["pixel_bboxes"], 16)):
        diff = np.abs(got - ref)
        bad = np.flatnonzero(diff.any(axis=1))
        print(f"\ngarden {name}: {len(bad)} of {N} gaussians differ from the oracle (max |diff| {int(diff.max())})")
        assert len(bad) <= 2e-5 * N, len(bad)
        assert diff.max() <= step                                  # never by more than one tile


@pytest.fixture(scope="module")
def box4k(G):
    n = int(os.environ.get("GSR_TEST_BOX_N", "20000000"))
    packed = G.utils.pack_gaussians(G.synthetic.uniform_box(n, 20))
    W, H = 3840, 2160
    p = G.synthetic.box_camera()
    fx = G.synthetic.pinhole_focal(W)
    args = (p.qvec, p.tvec, 2 * fx, 2 * fx, 2 * W, 2 * H, W, H)
    scene = G.renderer.GaussianScene.from_packed(packed)
    return packed, scene, G.renderer.make_camera(*args), G.orc.camera(*args)


def test_configs4_box_4k_vs_oracle(G, box4k):
    packed, scene, cam, ocam = box4k
    R = G.renderer.Rasterizer(scene)
    img = R.render(cam).cpu().numpy()
    st = dict(R.last_stats)
    oimg, _, drawn = _oracle_frame(G, packed, ocam)
    assert 0 < st["n_visible"] <= drawn and st["overflow"] == 0
    assert_frames_close(img, oimg)
    print(f"\nbox4k: {psnr(img, oimg):.1f} dB vs oracle, stats {st}")


def test_configs4_box_4k_exactness_properties(G, box4k):
    """At configs[4]'s size: footprint culling changes no bit, and 8 interleaved tile-row shards reassemble bit-exactly."""
    _, scene, cam, _ = box4k
    mk = G.renderer.make_options
    R = G.renderer.Rasterizer(scene)
    full, T = R.render(cam, return_T=True)
    culled_pairs = R.last_stats["n_pairs"]
    b, Tb = R.render(cam, mk(no_footprint_cull=True), return_T=True)
    assert torch.equal(full, b) and torch.equal(T, Tb)
    assert culled_pairs < R.last_stats["n_pairs"]
    del b, Tb
    tiles_y = (cam.height + 15) // 16
    out = torch.zeros_like(full)
    Rs = G.renderer.Rasterizer(scene, max_pairs=max(1 << 20, R.max_pairs // 4))
    del R
    for r in range(8):
        strip = Rs.render(cam, mk(tile_row_begin=r, tile_row_step=8, output_layout=2))
        k = len(range(r, tiles_y, 8))
        out.view(tiles_y, 16, cam.width, 3)[r::8] = strip[: k * 16].view(k, 16, cam.width, 3)
    assert torch.equal(out, full)


def _run_dist_check(backend, ranks, extra_env=None, timeout=600):
    env = dict(os.environ)
    env["GSR_BENCH_BACKEND"] = backend
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tools", "dist_check.py")]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_configs3_sharded_frame_over_gloo_ranks_sharing_the_gpu(G):
    """The N>1 path of bench.py (shard options, double-buffered gather, de-interleave) with real GPU strips: 3 ranks on this
    box's GPU, strips staged through the host for gloo.  Rank 0 compares the gathered frame with its own unsharded render."""
    p = _run_dist_check("gloo", 3)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_CHECK_OK" in p.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs for RCCL")
def test_configs3_sharded_frame_over_rccl(G):
    """One rank per GPU, RCCL gather over xGMI: frames bit-identical to the single-GPU frame, for several frames in a row
    (pins the buffer-reuse ordering between gather_async on the communicator stream and the next frame's render)."""
    ranks = min(torch.cuda.device_count(), 4)  # + this pytest process: within the box's limit of 6 GPU processes
    p = _run_dist_check("nccl", ranks)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST_CHECK_OK" in p.stdout
