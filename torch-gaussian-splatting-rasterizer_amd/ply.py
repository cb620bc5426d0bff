"""Minimal binary PLY reader/writer for INRIA-format trained gaussians.

Stands in for the third-party `plyfile.PlyData.read` the reference calls at
rasterize.py:353 (un-pinned in requirements.txt:5, not installed here).  The
reference only ever touches `plydata.elements[0][<column name>]`
(rasterize.py:98-106,355,358; utils.py:21,27), so `PlyData` below exposes exactly
that: `.elements` is a list whose items behave like `dict[str, np.ndarray]`.
Supports `binary_little_endian` / `binary_big_endian` / `ascii` with scalar
properties (list properties are rejected: gaussians files never have them).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1",
    "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


class PlyElement:
    """One PLY element: a named table of equally long columns."""

    def __init__(self, name: str, data: np.ndarray):
        self.name = name
        self.data = data  # structured array

    def __getitem__(self, key: str) -> np.ndarray:
        return self.data[key]

    def __contains__(self, key: str) -> bool:
        return key in (self.data.dtype.names or ())

    def __len__(self) -> int:
        return len(self.data)

    @property
    def properties(self) -> List[str]:
        return list(self.data.dtype.names or ())

    @staticmethod
    def describe(columns: Dict[str, np.ndarray], name: str = "vertex") -> "PlyElement":
        names = list(columns)
        n = len(columns[names[0]]) if names else 0
        rec = np.empty(n, dtype=[(k, np.asarray(columns[k]).dtype.newbyteorder("<")) for k in names])
        for k in names:
            rec[k] = columns[k]
        return PlyElement(name, rec)


class PlyData:
    def __init__(self, elements: List[PlyElement]):
        self.elements = elements

    def __getitem__(self, name: str) -> PlyElement:
        for el in self.elements:
            if el.name == name:
                return el
        raise KeyError(name)

    @staticmethod
    def read(path: str) -> "PlyData":
        with open(path, "rb") as fid:
            if fid.readline().strip() != b"ply":
                raise ValueError(f"{path}: not a PLY file")
            fmt = None
            layout = []  # [(element name, count, [(prop, dtype)])]
            while True:
                line = fid.readline()
                if not line:
                    raise ValueError(f"{path}: header has no end_header")
                tok = line.decode("ascii", "replace").split()
                if not tok or tok[0] in ("comment", "obj_info"):
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    layout.append((tok[1], int(tok[2]), []))
                elif tok[0] == "property":
                    if tok[1] == "list":
                        raise ValueError(f"{path}: list properties are not supported")
                    layout[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
                elif tok[0] == "end_header":
                    break
            if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
                raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
            elements = []
            for name, count, props in layout:
                if fmt == "ascii":
                    dtype = np.dtype([(p, "<" + t) for p, t in props])
                    rows = [fid.readline().split() for _ in range(count)]
                    data = np.empty(count, dtype=dtype)
                    for j, (p, _) in enumerate(props):
                        data[p] = [r[j] for r in rows]
                else:
                    order = "<" if fmt == "binary_little_endian" else ">"
                    dtype = np.dtype([(p, order + t) for p, t in props])
                    raw = fid.read(dtype.itemsize * count)
                    if len(raw) != dtype.itemsize * count:
                        raise EOFError(f"{path}: element {name!r} is truncated")
                    data = np.frombuffer(raw, dtype=dtype, count=count)
                elements.append(PlyElement(name, data))
        return PlyData(elements)

    def write(self, path: str) -> None:
        with open(path, "wb") as fid:
            head = ["ply", "format binary_little_endian 1.0"]
            for el in self.elements:
                head.append(f"element {el.name} {len(el)}")
                for p in el.properties:
                    if not p or any(ch.isspace() for ch in p):
                        raise ValueError(f"PLY property names are whitespace-delimited header tokens: {p!r} cannot be written")
                    kind = el.data.dtype[p]
                    ply_t = {"f4": "float", "f8": "double", "i4": "int", "u1": "uchar", "i2": "short",
                             "u2": "ushort", "u4": "uint", "i1": "char"}[kind.str[1:]]
                    head.append(f"property {ply_t} {p}")
            head.append("end_header")
            fid.write(("\n".join(head) + "\n").encode("ascii"))
            for el in self.elements:
                fid.write(np.ascontiguousarray(el.data).tobytes())


def write_gaussians_ply(path: str, columns: Dict[str, np.ndarray]) -> None:
    PlyData([PlyElement.describe({k: np.asarray(v, dtype=np.float32) for k, v in columns.items()})]).write(path)


def read_gaussians_columns(path: str) -> Dict[str, np.ndarray]:
    el = PlyData.read(path).elements[0]
    return {p: np.ascontiguousarray(el[p], dtype=np.float32) for p in el.properties}
