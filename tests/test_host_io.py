"""CPU: host-side IO and set-up logic — PLY / COLMAP round trips, SH packing, camera helper maths."""
import os

import numpy as np
import pytest

from conftest import golden_columns, load_golden


def test_ply_round_trip(tmp_path):
    from gsr_amd import ply, synthetic

    cols = synthetic.mip360_like(257, 3)
    path = str(tmp_path / "pc.ply")
    ply.write_gaussians_ply(path, cols)
    data = ply.PlyData.read(path)
    el = data.elements[0]
    assert len(el) == 257 and set(el.properties) == set(cols)
    for k, v in cols.items():
        assert np.array_equal(el[k], v)
    assert el.name == "vertex" and "opacity" in el and data["vertex"] is el


def test_ply_big_endian_and_ascii(tmp_path):
    from gsr_amd import ply

    vals = np.arange(12, dtype=np.float32).reshape(4, 3)
    head = "ply\nformat {} 1.0\ncomment x\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
    p1 = str(tmp_path / "be.ply")
    with open(p1, "wb") as f:
        f.write(head.format("binary_big_endian").encode() + vals.astype(">f4").tobytes())
    p2 = str(tmp_path / "as.ply")
    with open(p2, "w") as f:
        f.write(head.format("ascii") + "\n".join(" ".join(str(float(v)) for v in row) for row in vals) + "\n")
    for p in (p1, p2):
        el = ply.PlyData.read(p).elements[0]
        assert np.array_equal(np.stack([el["x"], el["y"], el["z"]], 1).astype(np.float32), vals)
    with pytest.raises(ValueError):
        bad = str(tmp_path / "bad.ply")
        open(bad, "wb").write(b"not a ply\n")
        ply.PlyData.read(bad)


def test_colmap_round_trip(tmp_path):
    from gsr_amd import data_reader as dr
    from gsr_amd import synthetic, utils

    poses = synthetic.ring_cameras(7, first_id=11)
    sparse = tmp_path / "sparse" / "0"
    os.makedirs(sparse)
    dr.write_extrinsics_binary(str(sparse / "images.bin"), poses)
    dr.write_intrinsics_binary(str(sparse / "cameras.bin"), [
        dr.Camera(id=1, model="PINHOLE", width=1920, height=1080, params=np.array([1000.0, 1001.0, 960.0, 540.0])),
        dr.Camera(id=5, model="SIMPLE_RADIAL", width=10, height=20, params=np.array([1.0, 2.0, 3.0, 4.0]))])
    images, cams = utils.read_scene(str(tmp_path))
    assert sorted(images) == list(range(11, 18))                       # keyed by COLMAP image_id (Q4)
    for p in poses:
        im = images[p.image_id]
        assert np.array_equal(im.qvec, p.qvec) and np.array_equal(im.tvec, p.tvec) and im.name == p.name and im.camera_id == 1
        assert im.xys.shape == (0, 2) and im.point3D_ids.shape == (0,)
    assert cams[1].model == "PINHOLE" and cams[1].width == 1920 and np.array_equal(cams[1].params, [1000.0, 1001.0, 960.0, 540.0])
    assert cams[5].model == "SIMPLE_RADIAL" and len(cams[5].params) == 4


def test_colmap_points2d_and_truncation(tmp_path):
    from gsr_amd import data_reader as dr

    class Im:
        id, qvec, tvec, camera_id, name = 3, [1, 0, 0, 0], [0, 0, 0], 1, "a.jpg"
        xys = np.array([[1.5, 2.5], [3.0, 4.0]])
        point3D_ids = np.array([7, -1])

    path = str(tmp_path / "images.bin")
    dr.write_extrinsics_binary(path, [Im])
    im = dr.read_extrinsics_binary(path)[3]
    assert np.array_equal(im.xys, Im.xys) and np.array_equal(im.point3D_ids, Im.point3D_ids)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-5])
    with pytest.raises(EOFError):
        dr.read_extrinsics_binary(path)


def test_pack_gaussians_layout():
    from gsr_amd import utils

    g = load_golden("f1_unit.npz")
    cols = golden_columns(g)
    p = utils.pack_gaussians(cols)
    n = len(cols["x"])
    assert p["means"].shape == (n, 3) and p["quats"].shape == (n, 4) and p["sh"].shape == (n, 16, 3)
    assert np.array_equal(p["quats"][:, 0], cols["rot_0"]) and np.array_equal(p["log_scales"][:, 2], cols["scale_2"])
    assert np.array_equal(p["sh"][:, 0, 1], cols["f_dc_1"]) and np.array_equal(p["sh"][:, 3, 2], cols["f_rest_32"])
    assert np.array_equal(p["sh"], g["sh"])
    for v in p.values():
        assert v.dtype == np.float32 and v.flags.c_contiguous


def test_synthetic_generators_are_deterministic():
    from gsr_amd import synthetic

    a, b = synthetic.mip360_like(1000, 9), synthetic.mip360_like(1000, 9)
    assert all(np.array_equal(a[k], b[k]) for k in a) and set(a) == set(synthetic.PLY_COLUMNS)
    c = synthetic.uniform_box(500, 2)
    assert np.abs(c["x"]).max() <= 10 and c["x"].dtype == np.float32
    for p in synthetic.ring_cameras(5) + [synthetic.box_camera()]:
        assert abs(np.linalg.norm(p.qvec) - 1) < 1e-12


def test_look_at_pose_geometry():
    """COLMAP convention x_cam = R(qvec) x_world + t: the eye maps to the origin, the target lies on +z,
    world-up maps to -y (y points down)."""
    from gsr_amd import synthetic

    def rot(q):
        w, x, y, z = q
        return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                         [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                         [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])

    for eye, target in (((4.0, 0.0, 1.0), (0, 0, 0)), ((0.5, -3.0, 1.0), (0.0, 0.0, 3.0)), ((-2.0, 2.5, 0.3), (1.0, 0.0, 0.2))):
        p = synthetic.look_at_pose(eye, target, 1, "a.png")
        R = rot(p.qvec)
        np.testing.assert_allclose(R @ np.array(eye) + p.tvec, 0, atol=1e-12)
        t_cam = R @ np.array(target, float) + p.tvec
        np.testing.assert_allclose(t_cam[:2], 0, atol=1e-12)
        assert t_cam[2] > 0
        assert (R @ np.array([0.0, 0.0, 1.0]))[1] < 0
    for p in synthetic.ring_cameras(8):
        c = rot(p.qvec) @ np.zeros(3) + p.tvec                     # the origin, seen from the ring camera
        np.testing.assert_allclose(c[:2], 0, atol=1e-9)
        assert abs(c[2] - np.hypot(4.0, 1.0)) < 1e-9


def test_morton_order_is_a_locality_preserving_permutation():
    """renderer.morton_order (GaussianScene spatial_order=True): a permutation of the gaussians along a Z-order curve of their
    means, balanced by rank quantisation whatever the scene's extent — consecutive gaussians are spatial neighbours."""
    from gsr_amd import renderer, synthetic

    cols = synthetic.mip360_like(20000, 9)
    m = np.stack([cols[k] for k in "xyz"], 1)
    p = renderer.morton_order(m)
    assert sorted(p.tolist()) == list(range(len(m))) and len(renderer.morton_order(m[:0])) == 0
    # neighbours in curve order are much closer in rank space (every axis rank-scaled to [0, 1]) than neighbours in file order
    r = np.stack([np.argsort(np.argsort(m[:, a])) for a in range(3)], 1) / len(m)
    step = lambda q: float(np.median(np.linalg.norm(np.diff(q, axis=0), axis=1)))
    assert step(r[p]) < 0.2 * step(r)
    assert np.array_equal(renderer.morton_order(m), p)          # deterministic
    # the loaders' path (torch sorts on the device the means live on; here the CPU) builds the same permutation, also with
    # repeated coordinates (stable sorts on both sides)
    import torch
    assert np.array_equal(renderer.morton_order_device(torch.from_numpy(m)).numpy(), p)
    dup = np.ascontiguousarray(m[np.arange(len(m)) % 97])
    assert np.array_equal(renderer.morton_order_device(torch.from_numpy(dup)).numpy(), renderer.morton_order(dup))
    assert len(renderer.morton_order_device(torch.from_numpy(m[:0]))) == 0


def test_inria_trained_model_header_is_read_by_name(tmp_path):
    """The file layout of a real trained model (62 float properties incl. the unused normals, INRIA order), written byte by byte by
    the test itself (conftest.write_inria_ply), not by ply.write_gaussians_ply: PlyData.read must find every column the reference
    reads BY NAME (rasterize.py:98-106,355,358; utils.py:21,27) among the extras, and pack_gaussians must lay them out as
    read_color_components does (sh[n,0,c] = f_dc_c, sh[n,k,c] = f_rest_{15c+k-1})."""
    from conftest import INRIA_PROPERTIES, golden_columns, load_golden, write_inria_ply
    from gsr_amd import ply, utils

    cols = golden_columns(load_golden("f1_unit.npz"))
    path = str(tmp_path / "point_cloud.ply")
    write_inria_ply(path, cols)
    assert os.path.getsize(path) == len(open(path, "rb").read().split(b"end_header\n")[0]) + len(b"end_header\n") + 62 * 4 * len(cols["x"])
    el = ply.PlyData.read(path).elements[0]
    assert el.name == "vertex" and el.properties == INRIA_PROPERTIES and len(el) == len(cols["x"])
    for k, v in cols.items():
        assert np.array_equal(el[k], v), k
    for k in ("nx", "ny", "nz"):
        assert not el[k].any()
    a, b = utils.pack_gaussians(el), utils.pack_gaussians(cols)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["sh"].shape == (len(cols["x"]), 16, 3) and np.array_equal(a["sh"][:, 0, 1], cols["f_dc_1"]) and np.array_equal(a["sh"][:, 3, 2], cols["f_rest_32"])
    got = ply.read_gaussians_columns(path)
    assert set(got) == set(INRIA_PROPERTIES) and all(v.dtype == np.float32 and v.flags["C_CONTIGUOUS"] for v in got.values())
