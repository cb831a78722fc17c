"""Shared scene builders / metrics for the parity tests."""
import math

import numpy as np
import torch

from ggsplat import synthetic as S

REL_L1_TOL = 1e-4        # BASELINE.json north_star: <= 1e-4 relative L1 on renders and gradients


def rel_l1(a, b) -> float:
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a.detach().cpu()).double()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b.detach().cpu()).double()
    return float((a - b).abs().sum() / (b.abs().sum() + 1e-30))


def quat_wxyz_to_rotmat(q):
    """Rotation matrix of (w, x, y, z) quaternions, in float64 (q and -q give the same matrix: sign-free comparisons)."""
    q = torch.as_tensor(q).detach().cpu().double()
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


def cam_kwargs(cam, bg):
    return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                bg=torch.as_tensor(bg, dtype=torch.float32), W=cam.image_width, H=cam.image_height,
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5))


def small_scene(P=600, W=80, H=64, sh_degree=3, seed=3, scale_mul=4.0, cam_index=1, opacity_boost=0.0):
    sc = S.random_gaussians(P, sh_degree=sh_degree, seed=seed)
    sc["scales"] = sc["scales"] * scale_mul
    if opacity_boost:
        g = torch.Generator().manual_seed(seed + 1)
        sc["opacities"] = torch.sigmoid(torch.randn(P, 1, generator=g) * 2 + opacity_boost)
    cam = S.orbit_cameras(4, width=W, img_height=H, fx=1.1 * W, fy=1.1 * W, cx=W / 2 - 2.0, cy=H / 2 + 1.0)[cam_index]
    return sc, cam


def seeded_image_weights(W, H, seed=11):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g) * 0.3,
            torch.randn(1, H, W, generator=g))


def poison_lds():
    """Leave NaNs in the LDS of (most of) the chip: sort / scan kernels of NaN-filled tensors stage their data there, and the
    LDS is not cleared between kernels.  A kernel of this repo that multiplies a stale LDS word by a zero weight shows up as
    NaN in its outputs -- only then, so run this right before the launch under test."""
    import torch
    x = torch.full((1024, 2048), float("nan"), device="cuda")
    torch.sort(x, dim=1)
    torch.cumsum(x, dim=1)
    torch.cuda.synchronize()
