"""Gaussian-model persistence in the reference's exact PLY layout.

The reference writes, through `plyfile`, one binary little-endian `vertex` element whose float32 properties
are, in this order (scene/gaussian_model.py:179-209):
    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3(K-1)-1)  opacity  scale_0..2  rot_0..3
with the SH blocks stored channel-major (`_features_*.transpose(1, 2).flatten(1)`), normals zero, and all
values pre-activation.  `plyfile` is not installable here, and it is not needed: the format is a text header
followed by packed float32 records, read / written below with numpy only.  Files are interchangeable with the
reference's `save_ply` / `load_ply` (:179-259) and `MeshGaussianModel.save_ply` (local_point_cloud.ply).
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np
import torch


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """construct_list_of_attributes (scene/gaussian_model.py:179-191)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def write_ply(path: str, columns: Dict[str, np.ndarray], order: List[str]) -> None:
    n = len(next(iter(columns.values())))
    rec = np.empty(n, dtype=[(k, "<f4") for k in order])
    for k in order:
        rec[k] = np.asarray(columns[k], dtype=np.float32).reshape(n)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {k}\n" for k in order) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply(path: str) -> Dict[str, np.ndarray]:
    """First `vertex` element of a binary-little-endian or ascii PLY -> {property: array}."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    assert lines[0].strip() == "ply", "not a PLY file"
    fmt, n, props, in_vertex = None, 0, [], False
    for ln in lines[1:]:
        t = ln.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            in_vertex = t[1] == "vertex" and not props
            if in_vertex:
                n = int(t[2])
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError("list properties are not supported in the vertex element")
            props.append((t[2], _PLY_TYPES[t[1]]))
    if fmt == "binary_little_endian":
        rec = np.frombuffer(data, dtype=np.dtype(props), count=n, offset=end)
    elif fmt == "ascii":
        rows = np.loadtxt(data[end:].decode("ascii").splitlines()[:n], ndmin=2)
        rec = np.empty(n, dtype=np.dtype(props))
        for i, (k, _) in enumerate(props):
            rec[k] = rows[:, i]
    else:
        raise ValueError(f"unsupported PLY format {fmt}")
    return {k: np.array(rec[k]) for k, _ in props}


def save_gaussians(path: str, _xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation) -> None:
    """GaussianModel.save_ply (scene/gaussian_model.py:193-209): pre-activation values, zero normals."""
    c = lambda t: t.detach().cpu().float()
    xyz = c(_xyz).numpy()
    f_dc = c(_features_dc).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = c(_features_rest).transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    op, sc, rot = c(_opacity).numpy(), c(_scaling).numpy(), c(_rotation).numpy()
    order = attribute_names(f_dc.shape[1], f_rest.shape[1], sc.shape[1], rot.shape[1])
    cols = {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]}
    cols.update({k: np.zeros(len(xyz), np.float32) for k in ("nx", "ny", "nz")})
    cols.update({f"f_dc_{i}": f_dc[:, i] for i in range(f_dc.shape[1])})
    cols.update({f"f_rest_{i}": f_rest[:, i] for i in range(f_rest.shape[1])})
    cols["opacity"] = op[:, 0]
    cols.update({f"scale_{i}": sc[:, i] for i in range(sc.shape[1])})
    cols.update({f"rot_{i}": rot[:, i] for i in range(rot.shape[1])})
    write_ply(path, cols, order)


def load_gaussians(path: str, max_sh_degree: int, device="cuda") -> Dict[str, torch.Tensor]:
    """GaussianModel.load_ply (scene/gaussian_model.py:216-259) -> parameter tensors in the model's layout
    (_features_dc [P,1,3], _features_rest [P,K-1,3])."""
    d = read_ply(path)
    P = len(d["x"])
    K = (max_sh_degree + 1) ** 2
    xyz = np.stack([d["x"], d["y"], d["z"]], 1)
    f_dc = np.stack([d["f_dc_0"], d["f_dc_1"], d["f_dc_2"]], 1).reshape(P, 3, 1)
    rest_names = sorted([k for k in d if k.startswith("f_rest_")], key=lambda s: int(s.split("_")[-1]))
    assert len(rest_names) == 3 * K - 3, f"PLY holds {len(rest_names)} f_rest values, sh degree {max_sh_degree} needs {3 * K - 3}"
    f_rest = np.stack([d[k] for k in rest_names], 1).reshape(P, 3, K - 1) if rest_names else np.zeros((P, 3, 0), np.float32)
    sc = np.stack([d[k] for k in sorted([k for k in d if k.startswith("scale_")], key=lambda s: int(s.split("_")[-1]))], 1)
    rot = np.stack([d[k] for k in sorted([k for k in d if k.startswith("rot")], key=lambda s: int(s.split("_")[-1]))], 1)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return {"_xyz": t(xyz), "_features_dc": t(f_dc).transpose(1, 2).contiguous(),
            "_features_rest": t(f_rest).transpose(1, 2).contiguous(), "_opacity": t(d["opacity"].reshape(P, 1)),
            "_scaling": t(sc), "_rotation": t(rot)}
