"""Mesh / binding persistence of a stage-2 output directory.

A registered frame of the reference lives on disk as
    <out>/stage2/<sequence>/point_cloud/frame_00012/local_point_cloud.ply   mesh-frame Gaussian parameters
    <out>/stage2/<sequence>/point_cloud/frame_00012/binding.pkl             pickled LongTensor: face of each Gaussian
    <out>/stage2/<sequence>/point_cloud/frame_00012/point_cloud.ply         world-frame parameters
    <out>/stage2/<sequence>/meshes/frame_00012.obj                          the registered garment mesh
(scene/scene.py:183-192, scene/mesh_gaussian_model.py:251-283 and :438-441).  This module reads and writes the OBJ
dialect of utils/io_utils.py:7-60 -- `v x y z`, optional `vt u v`, faces `f a b c` or `f a/ta b/tb c/tc`, 1-based --
and the binding pickle.  Text formats, so the files are interchangeable with the reference's in both directions.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional

import numpy as np
import torch


def write_obj(mesh: Dict[str, np.ndarray], filename: str) -> None:
    """Keys used (all optional): vertices [V,3], uvs [Vt,2], faces [F,3] 0-based, texture_faces [F,3] 0-based.
    Numbers are written with Python's shortest round-trip repr of the array's scalar type, like the reference's
    f-strings do, so a float32 vertex survives write -> read bit for bit."""
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    out = []
    if "vertices" in mesh:
        out += ["v %s %s %s\n" % (x, y, z) for x, y, z in np.asarray(mesh["vertices"])]
    if "uvs" in mesh:
        out += ["vt %s %s\n" % (u, v) for u, v in np.asarray(mesh["uvs"])]
    if "faces" in mesh:
        faces = np.asarray(mesh["faces"]).astype(np.int64) + 1
        if "texture_faces" in mesh:
            tf = np.asarray(mesh["texture_faces"]).astype(np.int64) + 1
            out += ["f %d/%d %d/%d %d/%d\n" % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(faces, tf)]
        else:
            out += ["f %d %d %d\n" % (a[0], a[1], a[2]) for a in faces]
    with open(filename, "w") as f:
        f.writelines(out)


def read_obj(filename: str) -> Dict[str, np.ndarray]:
    """-> {vertices f32 [V,3], uvs f32 [Vt,2], faces int [F,3] 0-based, texture_faces int [F,3] 0-based}.
    As in the reference, the texture index of a face corner is read only once a `vt` line has been seen, and a corner
    may be `a`, `a/t` or `a/t/n` (normals ignored).  Empty sections come back as empty arrays."""
    verts, uvs, faces, tfaces = [], [], [], []
    seen_vt = False
    with open(filename, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            tag = tok[0]
            if tag == "v":
                verts.append([float(t) for t in tok[1:]])
            elif tag == "vt":
                seen_vt = True
                uvs.append([float(t) for t in tok[1:]])
            elif tag == "f":
                corners = [t.split("/") for t in tok[1:]]
                faces.append([int(c[0]) for c in corners])
                if seen_vt:
                    tfaces.append([int(c[1]) for c in corners])
    return {"vertices": np.array(verts, dtype=np.float32), "uvs": np.array(uvs, dtype=np.float32),
            "faces": np.array(faces) - 1, "texture_faces": np.array(tfaces) - 1}


def save_binding(path: str, binding: torch.Tensor) -> None:
    """binding.pkl next to local_point_cloud.ply: `pickle.dump` of the tensor itself
    (scene/mesh_gaussian_model.py:277-282)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(binding, f)


def load_binding(path: str, device="cuda") -> torch.Tensor:
    """scene/mesh_gaussian_model.py:335-337.  The pickle holds whatever the writer had (a CUDA or CPU LongTensor; a
    numpy array or list from other tools is accepted too); returned as a contiguous int64 tensor on `device`."""
    with open(path, "rb") as f:
        b = pickle.load(f)
    return torch.as_tensor(np.asarray(b.cpu()) if torch.is_tensor(b) else np.asarray(b)).long().to(device).contiguous()


def frame_paths(stage2_sequence_dir: str, frame: int) -> Dict[str, str]:
    """Where scene/scene.py:183-192 and s2_registration.py put the files of one registered frame."""
    pc = os.path.join(stage2_sequence_dir, "point_cloud", f"frame_{int(frame):05d}")
    return {"local_ply": os.path.join(pc, "local_point_cloud.ply"), "binding": os.path.join(pc, "binding.pkl"),
            "world_ply": os.path.join(pc, "point_cloud.ply"),
            "mesh": os.path.join(stage2_sequence_dir, "meshes", f"frame_{int(frame):05d}.obj")}


def mesh_vertices(path: str, device="cuda") -> Optional[torch.Tensor]:
    """The `torch.tensor(read_obj(_mesh)['vertices'], device='cuda')` of scene/scene.py:155-156."""
    return torch.from_numpy(read_obj(path)["vertices"]).to(device)
