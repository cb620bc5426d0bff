"""COLMAP sparse-model binary readers (and the writers the tests/tools need).

Replaces the two readers the reference's render path uses
(reference data_reader.py:140-173 `read_extrinsics_binary`, :176-198
`read_intrinsics_binary`; called from utils.py:53,56).  Record formats:

  images.bin : u64 count; per image  i32 image_id, 4 f64 qvec (w,x,y,z), 3 f64 tvec,
               i32 camera_id, NUL-terminated name, u64 n_points2D, n * (f64 x, f64 y, i64 id)
  cameras.bin: u64 count; per camera i32 camera_id, i32 model_id, u64 width, u64 height,
               num_params(model_id) f64

The other COLMAP readers of the reference (text models, points3D, dense arrays;
data_reader.py:48-137,201-254) are never reached from the render path and are
out of scope.
"""
from __future__ import annotations

import collections
import struct
from typing import BinaryIO, Dict

import numpy as np

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
BaseImage = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])

# model_id -> (name, number of f64 params); COLMAP's camera model table
_MODELS = {
    0: ("SIMPLE_PINHOLE", 3),
    1: ("PINHOLE", 4),
    2: ("SIMPLE_RADIAL", 4),
    3: ("RADIAL", 5),
    4: ("OPENCV", 8),
    5: ("OPENCV_FISHEYE", 8),
    6: ("FULL_OPENCV", 12),
    7: ("FOV", 5),
    8: ("SIMPLE_RADIAL_FISHEYE", 4),
    9: ("RADIAL_FISHEYE", 5),
    10: ("THIN_PRISM_FISHEYE", 12),
}
_MODEL_IDS = {name: mid for mid, (name, _) in _MODELS.items()}

_IMG_HEAD = struct.Struct("<i7di")  # 64 bytes
_CAM_HEAD = struct.Struct("<iiQQ")  # 24 bytes
_POINT2D = np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")])


def _take(fid: BinaryIO, nbytes: int) -> bytes:
    buf = fid.read(nbytes)
    if len(buf) != nbytes:
        raise EOFError(f"truncated COLMAP file: wanted {nbytes} bytes, got {len(buf)}")
    return buf


def _read_cstring(fid: BinaryIO) -> str:
    out = bytearray()
    while True:
        ch = _take(fid, 1)
        if ch == b"\x00":
            return out.decode("utf-8")
        out += ch


def read_extrinsics_binary(path_to_model_file: str) -> Dict[int, BaseImage]:
    """`images.bin` -> {image_id: Image}.  Keyed by COLMAP image_id (quirk Q4)."""
    images: Dict[int, BaseImage] = {}
    with open(path_to_model_file, "rb") as fid:
        (count,) = struct.unpack("<Q", _take(fid, 8))
        for _ in range(count):
            head = _IMG_HEAD.unpack(_take(fid, _IMG_HEAD.size))
            image_id, camera_id = head[0], head[8]
            name = _read_cstring(fid)
            (n2d,) = struct.unpack("<Q", _take(fid, 8))
            pts = np.frombuffer(_take(fid, 24 * n2d), dtype=_POINT2D, count=n2d)
            images[image_id] = BaseImage(
                id=image_id,
                qvec=np.array(head[1:5], dtype=np.float64),
                tvec=np.array(head[5:8], dtype=np.float64),
                camera_id=camera_id,
                name=name,
                xys=np.column_stack([pts["x"], pts["y"]]).astype(np.float64),
                point3D_ids=pts["id"].astype(np.int64),
            )
    return images


def read_intrinsics_binary(path_to_model_file: str) -> Dict[int, Camera]:
    """`cameras.bin` -> {camera_id: Camera}."""
    cameras: Dict[int, Camera] = {}
    with open(path_to_model_file, "rb") as fid:
        (count,) = struct.unpack("<Q", _take(fid, 8))
        for _ in range(count):
            camera_id, model_id, width, height = _CAM_HEAD.unpack(_take(fid, _CAM_HEAD.size))
            if model_id not in _MODELS:
                raise KeyError(model_id)
            model_name, n_params = _MODELS[model_id]
            params = np.frombuffer(_take(fid, 8 * n_params), dtype="<f8").astype(np.float64)
            cameras[camera_id] = Camera(id=camera_id, model=model_name, width=width, height=height, params=params)
    if len(cameras) != count:
        raise AssertionError("duplicate camera ids in cameras.bin")
    return cameras


# ---------------------------------------------------------------------------
# writers (tests, golden-vector tool, synthetic on-disk scenes)
# ---------------------------------------------------------------------------
def write_extrinsics_binary(path: str, images) -> None:
    """`images`: iterable of objects with id/image_id, qvec, tvec, camera_id (default 1), name."""
    images = list(images)
    with open(path, "wb") as fid:
        fid.write(struct.pack("<Q", len(images)))
        for im in images:
            image_id = getattr(im, "id", None)
            if image_id is None:
                image_id = im.image_id
            q = [float(v) for v in im.qvec]
            t = [float(v) for v in im.tvec]
            fid.write(_IMG_HEAD.pack(int(image_id), *q, *t, int(getattr(im, "camera_id", 1))))
            fid.write(im.name.encode("utf-8") + b"\x00")
            xys = getattr(im, "xys", None)
            n2d = 0 if xys is None else len(xys)
            fid.write(struct.pack("<Q", n2d))
            if n2d:
                pts = np.empty(n2d, dtype=_POINT2D)
                pts["x"], pts["y"] = np.asarray(xys)[:, 0], np.asarray(xys)[:, 1]
                pts["id"] = np.asarray(im.point3D_ids)
                fid.write(pts.tobytes())


def write_intrinsics_binary(path: str, cameras) -> None:
    cameras = list(cameras)
    with open(path, "wb") as fid:
        fid.write(struct.pack("<Q", len(cameras)))
        for cam in cameras:
            model_id = _MODEL_IDS[cam.model]
            params = np.asarray(cam.params, dtype="<f8")
            if params.size != _MODELS[model_id][1]:
                raise ValueError(f"{cam.model} takes {_MODELS[model_id][1]} params")
            fid.write(_CAM_HEAD.pack(int(cam.id), model_id, int(cam.width), int(cam.height)))
            fid.write(params.tobytes())
