"""Synthetic workloads of SURVEY.md section 8(d) (there is no dataset on the GPU box).

config 1: 10k free Gaussians, 4 cameras 512x512 on a radius-4 circle.
config 2: 100k mesh-bound Gaussians on an open "skirt" tube, 160 ActorsHQ-style
          1080p cameras (5 rings x 32 azimuths).
config 5: same generator at 500k faces / 4K.
All generators are seeded and device-agnostic (tensors are created on CPU and
moved by the caller).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from .cameras import Camera, look_at_camera

SH_C0 = 0.28209479177387814


def RGB2SH(rgb):
    """utils/sh_utils.py:113-114."""
    return (rgb - 0.5) / SH_C0


def random_gaussians(P: int = 10_000, sh_degree: int = 3, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Config 1 scene: activated (post-exp / post-sigmoid / normalised) rasterizer inputs."""
    g = torch.Generator().manual_seed(seed)
    K = (sh_degree + 1) ** 2
    means = torch.rand(P, 3, generator=g) * 2 - 1
    lo, hi = math.log(0.005), math.log(0.05)
    scales = torch.exp(torch.rand(P, 3, generator=g) * (hi - lo) + lo)
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5)
    shs = torch.randn(P, K, 3, generator=g) * 0.1
    shs[:, 0] = RGB2SH(torch.rand(P, 3, generator=g))
    return dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs, sh_degree=sh_degree)


def orbit_cameras(n: int = 4, radius: float = 4.0, height: float = 0.5, width: int = 512, img_height: int = 512,
                  fx: float = 600.0, fy: float = 600.0, cx: float = 250.0, cy: float = 262.0,
                  target=(0.0, 0.0, 0.0), device="cpu") -> List[Camera]:
    cams = []
    for i in range(n):
        a = 2 * math.pi * i / n
        eye = (radius * math.cos(a), height, radius * math.sin(a))
        cams.append(look_at_camera(eye, target, width=width, height=img_height, fx=fx, fy=fy, cx=cx, cy=cy,
                                   uid=i, device=device))
    return cams


def skirt_mesh(n_around: int = 200, n_rows: int = 250, r_top: float = 0.30, r_bottom: float = 0.50,
               height: float = 0.8, y_center: float = 1.0, jitter: float = 1e-3, seed: int = 0
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Open tube, wrapped around: V = n_around (n_rows+1), F = 2 n_around n_rows (100 000 at defaults)."""
    g = torch.Generator().manual_seed(seed)
    j = torch.arange(n_rows + 1, dtype=torch.float64) / n_rows          # 0 top .. 1 bottom
    i = torch.arange(n_around, dtype=torch.float64) / n_around * 2 * math.pi
    rad = r_top + (r_bottom - r_top) * j
    y = y_center + height / 2 - height * j
    X = rad[:, None] * torch.cos(i)[None, :]
    Z = rad[:, None] * torch.sin(i)[None, :]
    Y = y[:, None].expand_as(X)
    v = torch.stack([X, Y, Z], dim=-1).reshape(-1, 3).to(torch.float32)
    v = v + torch.randn(v.shape, generator=g) * jitter
    jj, ii = torch.meshgrid(torch.arange(n_rows), torch.arange(n_around), indexing="ij")
    v00 = jj * n_around + ii
    v10 = jj * n_around + (ii + 1) % n_around
    v01 = (jj + 1) * n_around + ii
    v11 = (jj + 1) * n_around + (ii + 1) % n_around
    f = torch.stack([torch.stack([v00, v10, v11], -1), torch.stack([v00, v11, v01], -1)], dim=2).reshape(-1, 3)
    return v, f.to(torch.int64)


def skirt_gaussian_params(F: int, sh_degree: int = 0, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Raw (pre-activation) MeshGaussianModel parameters, one Gaussian per face (binding = arange(F),
    scene/mesh_gaussian_model.py:82).  Local y is the face normal, so it is the thin axis."""
    g = torch.Generator().manual_seed(seed + 17)
    K = (sh_degree + 1) ** 2
    s = torch.stack([torch.rand(F, generator=g) * 0.4 + 0.4, torch.full((F,), 0.1),
                     torch.rand(F, generator=g) * 0.4 + 0.4], dim=-1)
    rot = torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(F, 1) + torch.randn(F, 4, generator=g) * 0.05
    feats = torch.randn(F, K, 3, generator=g) * 0.1
    feats[:, 0] = RGB2SH(torch.rand(F, 3, generator=g))
    return dict(_xyz=torch.zeros(F, 3), _scaling=torch.log(s), _rotation=rot,
                _opacity=torch.randn(F, 1, generator=g) + 2.0,
                _features_dc=feats[:, :1].contiguous(), _features_rest=feats[:, 1:].contiguous(),
                binding=torch.arange(F, dtype=torch.int64))


def rig_cameras(n_rings: int = 5, n_az: int = 32, radius: float = 2.5, h_lo: float = 0.2, h_hi: float = 2.2,
                width: int = 1920, height: int = 1080, f: float = 1500.0, target=(0.0, 1.0, 0.0), seed: int = 1,
                device="cpu") -> List[Camera]:
    """160 ActorsHQ-style cameras: principal point (W/2, H/2) + U[-8, 8]."""
    rng = np.random.default_rng(seed)
    cams = []
    for r in range(n_rings):
        hy = h_lo + (h_hi - h_lo) * r / max(n_rings - 1, 1)
        for k in range(n_az):
            a = 2 * math.pi * (k + 0.5 * (r % 2)) / n_az
            eye = (radius * math.cos(a), hy, radius * math.sin(a))
            cx = width / 2 + rng.uniform(-8, 8)
            cy = height / 2 + rng.uniform(-8, 8)
            cams.append(look_at_camera(eye, target, width=width, height=height, fx=f, fy=f, cx=cx, cy=cy,
                                       uid=r * n_az + k, device=device))
    return cams


def stack_cameras(cams: List[Camera], device=None) -> Dict[str, torch.Tensor]:
    """Pack per-view camera data for the batched entry points: view [V,16], proj [V,16], campos [V,3],
    tanfov [V,2] (row-major flat copies of the transposed matrices the reference passes)."""
    view = torch.stack([c.world_view_transform.reshape(16) for c in cams]).float()
    proj = torch.stack([c.full_proj_transform.reshape(16) for c in cams]).float()
    campos = torch.stack([c.camera_center.reshape(3) for c in cams]).float()
    tanfov = torch.tensor([[math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5)] for c in cams], dtype=torch.float32)
    out = dict(view=view, proj=proj, campos=campos, tanfov=tanfov)
    if device is not None:
        out = {k: v.to(device).contiguous() for k, v in out.items()}
    return out
