"""CPU: the C oracle (oracle/gsr_oracle.c) against golden vectors captured from the real reference
(tools/make_golden.py).  This is what pins the oracle; the GPU tests then compare HIP vs oracle."""
import numpy as np
import pytest

from conftest import golden_columns, load_golden, psnr
from oracle import cpu_oracle as orc

import gsr_amd
from gsr_amd import synthetic, utils


def _cam(g, prefix=""):
    return orc.camera(g[prefix + "qvec"], g[prefix + "tvec"], float(g["fx_full"]), float(g["fy_full"]),
                      int(g["cam_width"]), int(g["cam_height"]), int(g["width"]), int(g["height"]))


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_f1_camera_matrices():
    g = load_golden("f1_unit.npz")
    cam = _cam(g)
    w2c = np.array(cam.w2c, np.float32).reshape(4, 4)
    assert np.array_equal(w2c, g["w2c_T"])                      # bit-exact: rasterize.py:59-77,:361
    assert np.array_equal(w2c.T, g["w2c_M"])
    full_ref = (g["w2c_T"].astype(np.float32) @ g["proj_P"].T.astype(np.float32))
    full = np.array(cam.full_proj, np.float32).reshape(4, 4)
    np.testing.assert_allclose(full[:, [0, 1, 3]], full_ref[:, [0, 1, 3]], rtol=0, atol=0)
    np.testing.assert_allclose(full, full_ref, rtol=2e-7)
    assert np.float32(g["tan_fov_x"]) == np.float32(cam.tan_fov_x)
    assert np.float32(1.3 * float(g["tan_fov_x"])) == np.float32(cam.lim_x)
    assert np.float32(float(g["focals"][0]) / 2) == np.float32(cam.focal_x)
    inv = np.linalg.inv(g["w2c_T"].astype(np.float64))[3, :3]
    np.testing.assert_allclose(np.array(cam.cam_center), inv, rtol=1e-6, atol=1e-7)


def test_f1_sh_packing():
    g = load_golden("f1_unit.npz")
    assert np.array_equal(utils.sh_columns_to_array(golden_columns(g)), g["sh"])   # utils.py:10-31


def test_f1_intermediates():
    g = load_golden("f1_unit.npz")
    packed = utils.pack_gaussians(golden_columns(g))
    pre = orc.preprocess(packed, _cam(g))
    assert _rel(pre["cov3d"], g["cov3d"]) < 1e-6
    assert _rel(pre["cam_means"], g["cam_means"]) < 1e-6
    assert _rel(pre["rgb"], g["rgb"]) < 1e-6
    assert _rel(pre["opacity"], g["opacity"]) < 1e-6
    vis = g["cam_means"][:, 2] >= 0.2
    assert _rel(pre["cov2d"][vis], g["cov2d"][vis]) < 2e-5
    assert _rel(pre["screen_means"], g["screen_means"]) < 1e-5
    assert _rel(pre["sigmas"], g["sigmas"]) < 2e-4
    assert np.array_equal(pre["tile_bboxes"], g["tile_bboxes"])
    assert np.array_equal(pre["pixel_bboxes"], g["pixel_bboxes"])


@pytest.mark.parametrize("name,floor_db", [("f1_unit.npz", 120.0), ("f2_small.npz", 120.0)])
def test_image_matches_reference(name, floor_db):
    g = load_golden(name)
    packed = utils.pack_gaussians(golden_columns(g))
    cam = _cam(g)
    pre = orc.preprocess(packed, cam)
    order = orc.depth_order(pre["cam_means"])
    screen, T, drawn = orc.composite(order, pre, cam.width, cam.height)
    img = screen.transpose(1, 0, 2)
    assert drawn == len(g["draw_order"])
    # drawn order == the reference's (no depth ties in the fixture)
    mine = [int(i) for i in order if (np.prod(pre["pixel_bboxes"][i, 2:] - pre["pixel_bboxes"][i, :2]) != 0
                                       and np.all(pre["sigmas"][i] != 0))]
    assert mine == [int(i) for i in g["draw_order"]]
    assert psnr(img, g["image"]) >= floor_db
    assert np.abs(img - g["image"]).max() < 1e-5
    # Q1: last row / column stay black
    assert not img[-1].any() and not img[:, -1].any()
    # whole-frame entry point and threaded banding give the same bits
    img2, drawn2 = orc.render(packed, cam, threads=1)
    img3, _ = orc.render(packed, cam, threads=4)
    assert drawn2 == drawn and np.array_equal(img2, img) and np.array_equal(img3, img)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_f3_edge_cases(tag):
    g = load_golden("f3_edge.npz")
    packed = utils.pack_gaussians(golden_columns(g))
    cam = _cam(g, prefix=tag + "_")
    pre = orc.preprocess(packed, cam)
    assert np.array_equal(pre["pixel_bboxes"], g[tag + "_pixel_bboxes"])
    assert np.array_equal(pre["tile_bboxes"], g[tag + "_tile_bboxes"])
    assert _rel(pre["rgb"], g[tag + "_rgb"]) < 1e-6
    # Q2: exact zeros in the conic are reproduced exactly (they gate the skip at rasterize.py:441)
    assert np.array_equal(pre["sigmas"] == 0, g[tag + "_sigmas"] == 0)
    order = orc.depth_order(pre["cam_means"])
    screen, T, drawn = orc.composite(order, pre, cam.width, cam.height)
    img = screen.transpose(1, 0, 2)
    assert drawn == len(g[tag + "_draw_order"])
    assert psnr(img, g[tag + "_image"]) >= 110.0
    assert bool(g["q4_keyerror"])


def test_f3_q2_axis_aligned_gaussian_is_skipped():
    g = load_golden("f3_edge.npz")
    assert 0 not in set(int(i) for i in g["a_draw_order"])      # gaussian 0: on-axis, axis-aligned -> sigma_xy == 0
    assert g["a_sigmas"][0, 2] == 0 and g["a_sigmas"][0, 0] != 0


def test_f4_medium_regenerated_inputs():
    g = load_golden("f4_medium.npz")
    cols = synthetic.mip360_like(int(g["n"]), int(g["seed"]))
    for i in range(3):
        cols[f"scale_{i}"] = (cols[f"scale_{i}"] + np.float32(float(g["scale_shift"]))).astype(np.float32)
    checksum = sum(float(np.asarray(v, np.float64).sum()) for v in cols.values())
    assert checksum == float(g["ply_checksum"]), "synthetic generator drifted from the one that made the fixture"
    packed = utils.pack_gaussians(cols)
    img, drawn = orc.render(packed, _cam(g), threads=None)
    assert drawn == int(g["n_drawn"])
    y0, x0 = int(g["crop_y0"]), int(g["crop_x0"])
    assert psnr(img[y0:y0 + 256, x0:x0 + 256], g["crop"]) >= 110.0
    assert psnr(img[::4, ::4], g["decimated"]) >= 110.0
    np.testing.assert_allclose(img.astype(np.float64).sum(axis=(1, 2)), g["row_sums"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(img.astype(np.float64).sum(axis=(0, 2)), g["col_sums"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["f1_unit.npz", "f2_small.npz"])
def test_torch_loop_port_reproduces_the_reference_frames(name):
    """oracle/torch_loop.py is what bench.py times as cpu_baseline (kind "port"): the reference's per-gaussian loop
    (rasterize.py:436-446, :255-305) re-stated in torch ops.  Same frame as the reference's own, same draw count."""
    from oracle import torch_loop

    g = load_golden(name)
    cam = _cam(g)
    pre = orc.preprocess(utils.pack_gaussians(golden_columns(g)), cam)
    order = orc.depth_order(pre["cam_means"])
    img, drawn = torch_loop.render(pre, order, cam.width, cam.height)
    assert drawn == len(g["draw_order"])
    assert psnr(img, g["image"]) >= 120.0, psnr(img, g["image"])
    assert not img[-1].any() and not img[:, -1].any()           # Q1
    # and the timed-sample estimator walks the same two classes of iterations
    s = torch_loop.timed_sample(pre, order, cam.width, cam.height, budget_s=2.0, max_gaussians=500)
    assert s["total_drawn"] == drawn and s["total_iterations"] == len(order) and s["sampled"] <= 500


def test_f5_deep_stacks_oracle_vs_the_reference():
    """Fixture f5 = the reference's own frame of the fuzz case that needed the campaign's looser per-pixel threshold
    (case-seed 2519059510838425248: 45 918 gaussians, scales blown up e-fold, a 15x360 frame: hundreds of semi-transparent
    layers per pixel).  Round 2 left open whether the HIP path's residual on it (37 pixels 1e-5 .. 2e-5 from the oracle) was
    its log2-domain exponent.  It is not: the ORACLE (plain C, expf, the reference's own formula and order) differs from
    the reference's torch arithmetic by more than that — measured here: 192 of 5400 pixels off by more than 1e-5, the worst
    3.3e-5, 111.7 dB.  Deep stacks accumulate every alpha's last-ulp difference in T; the draw sets are identical."""
    import os
    import sys

    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import fuzz_parity

    g = load_golden("f5_deep_stack.npz")
    c = fuzz_parity.build_case(int(g["case_seed"]), int(g["max_n"]))
    assert (c["n"], c["W"], c["H"]) == (int(g["n"]), int(g["width"]), int(g["height"]))
    img, drawn = orc.render(c["packed"], orc.camera(*c["args"]))
    assert drawn == int(g["n_drawn"])
    d = np.abs(img.astype(np.float64) - g["image"]).max(axis=2)
    print(f"\nf5 oracle vs reference: {(d > 1e-5).sum()} of {d.size} pixels off by > 1e-5, worst {d.max():.2e}, {psnr(img, g['image']):.1f} dB")
    assert d.max() <= 1e-4 and psnr(img, g["image"]) >= 105.0
