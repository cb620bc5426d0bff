"""CPU property tests (hypothesis) for the host-side logic and the oracle's invariances."""
import os

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from conftest import golden_columns, load_golden


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 400), st.integers(1, 300), st.integers(1, 9), st.integers(1, 2))
def test_tile_row_plan_partitions_any_frame(H, W, world, block):
    from gsr_amd import dist as gdist, renderer

    plan = gdist.TileRowPlan(H, W, world, block)
    rows = sorted(r for rs in plan.rows for r in rs)
    assert rows == list(range((H + 15) // 16))
    for r in range(world):  # the plan's rows are what libgsr's options describe (renderer.shard_row_list restates RowShard::row_at)
        so = plan.shard_options(r)
        assert plan.rows[r] == renderer.shard_row_list(H, so["tile_row_begin"], so["tile_row_step"], so["tile_row_block"])
        if block == 2:
            assert all(t ^ 1 in plan.rows[r] or t ^ 1 >= plan.tiles_y for t in plan.rows[r])  # whole cell rows
    frame = torch.arange(H * W * 3, dtype=torch.float32).view(H, W, 3)
    strips = [plan.split(frame, r) for r in range(world)]
    assert all(tuple(s.shape) == plan.padded_shape() for s in strips)
    assert torch.equal(plan.assemble(strips), frame)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 60), st.integers(0, 2 ** 32 - 1))
def test_ply_round_trip_any_table(n, seed):
    import tempfile

    from gsr_amd import ply

    rng = np.random.default_rng(seed)
    cols = {k: rng.standard_normal(n).astype(np.float32) for k in ("x", "y", "z", "opacity", "f_dc_0", "weird-name_1")}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.ply")
        ply.write_gaussians_ply(path, cols)
        back = ply.read_gaussians_columns(path)
    assert list(back) == list(cols) and all(np.array_equal(back[k], cols[k]) for k in cols)


def test_ply_writer_rejects_names_the_format_cannot_hold(tmp_path):
    import pytest

    from gsr_amd import ply

    with pytest.raises(ValueError):
        ply.write_gaussians_ply(str(tmp_path / "bad.ply"), {"weird name": np.zeros(3, np.float32)})


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_oracle_frame_is_invariant_to_gaussian_order(seed):
    """permuting the input gaussians does not change the frame (no depth ties in the fixture)."""
    from gsr_amd import utils
    from oracle import cpu_oracle as orc

    g = load_golden("f1_unit.npz")
    cols = golden_columns(g)
    perm = np.random.default_rng(seed).permutation(len(cols["x"]))
    cam = orc.camera(g["qvec"], g["tvec"], float(g["fx_full"]), float(g["fy_full"]), int(g["cam_width"]), int(g["cam_height"]),
                     int(g["width"]), int(g["height"]))
    a, _ = orc.render(utils.pack_gaussians(cols), cam, threads=1)
    b, _ = orc.render(utils.pack_gaussians({k: v[perm] for k, v in cols.items()}), cam, threads=1)
    assert np.array_equal(a, b)


@settings(max_examples=30, deadline=None)
@given(st.floats(-3, 3), st.floats(-3, 3), st.floats(0.5, 6), st.floats(-3.1, 3.1), st.floats(-1.2, 1.2))
def test_camera_setup_places_the_target_on_the_optical_axis(ex, ey, dist, yaw, pitch):
    from gsr_amd import _lib, synthetic

    eye = np.array([ex, ey, 0.3])
    fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch) * 0.7])
    if np.linalg.norm(np.cross(fwd, [0, 0, 1])) < 1e-3:
        return
    target = eye + dist * fwd / np.linalg.norm(fwd)
    p = synthetic.look_at_pose(eye, target, 1, "a.png")
    cam = _lib.camera_setup(p.qvec, p.tvec, 800.0, 800.0, 640, 480, 320, 240)
    w2c = np.array(cam.w2c, np.float64).reshape(4, 4)
    t_cam = target @ w2c[:3, :3] + w2c[3, :3]                    # row-vector convention of the reference
    assert abs(t_cam[0]) < 1e-4 * max(1, dist) and abs(t_cam[1]) < 1e-4 * max(1, dist) and abs(t_cam[2] - dist) < 1e-4 * max(1, dist)
    np.testing.assert_allclose(np.array(cam.cam_center), eye, atol=1e-4)
