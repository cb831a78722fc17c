"""Dense, autograd-differentiable CPU restatement of the Gaussian-splat rasterizer.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may use it, and only as the checker.

PARITY UNPINNED: the arithmetic of this path lives in the third-party CUDA
extension ``diff_gaussian_rasterization_depth_alpha`` (lizhe00/AnimatableGaussians,
``gaussians/diff_gaussian_rasterization_depth_alpha``, cloned un-pinned by the
reference's ``setup.sh:26-28``).  Its source is not under /root/reference and the
reference holds no tests or golden vectors for it, so this file restates the
published algorithm (Kerbl et al. 2023, "3D Gaussian Splatting", tile rasterizer
with the depth/alpha outputs of that fork) as summarised in SURVEY.md Appendix A,
anchored on the reference's own call site ``gaussian_renderer/__init__.py:39-54,
103-111`` (argument meaning, return order ``(color, radii, depth, alpha)``).
The two stages that DO exist in the reference tree are pinned against it by
golden vectors (tests/golden): SH evaluation (``utils/sh_utils.py:56-111``) and
scale/rotation -> cov3D (``utils/general_utils.py:91-120``,
``scene/gaussian_model.py:27-31``).

Why a second oracle next to ``splat_oracle.c``: gradients here come from
autograd, so they check the hand-derived backward of the C restatement (and of
the HIP kernels) independently.  It is O(tile_list x 256) dense per tile: use it
for <= ~10k Gaussians / <= 512x512.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

# ---- named constants of the upstream rasterizer -------
TILE = 16                 # tile edge in pixels (A.0)
NEAR_Z = 0.2              # cull p_view.z <= 0.2 (A.1 step 1)
W_EPS = 1e-7              # p_hom.w + 1e-7 (A.1 step 2)
FOV_CLAMP = 1.3           # 1.3 * tanfov clamp of t.x/t.z (A.1 step 4)
LOWPASS = 0.3             # cov2D diagonal += 0.3 px^2 (A.1 step 4)
LAMBDA_FLOOR = 0.1        # sqrt(max(0.1, mid^2 - det)) (A.1 step 5)
RADIUS_SIGMA = 3.0        # radius = ceil(3 sqrt(lambda_max)) (A.1 step 5)
ALPHA_MAX = 0.99          # alpha = min(0.99, ...) (A.1 step 10)
ALPHA_MIN = 1.0 / 255.0   # skip alpha < 1/255 (A.1 step 10)
T_MIN = 1e-4              # stop when T (1 - alpha) < 1e-4 (A.1 step 10)

# SH constants: same values as the reference's utils/sh_utils.py:25-52
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def eval_sh_rgb(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH -> RGB (before the +0.5 / clamp).  ``sh`` is [P,K,3] (coefficient-major,
    RGB innermost: the rasterizer layout, A.0); ``dirs`` [P,3] unit.
    Follows utils/sh_utils.py:56-111 term for term (degrees 0..3)."""
    assert 0 <= deg <= 3
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> R, NOT normalised inside (the kernel assumes unit input, A.0).
    Same polynomial as utils/general_utils.py:101-109."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales: torch.Tensor, mod: float, rots: torch.Tensor) -> torch.Tensor:
    """Sigma = R S^2 R^T, stored (xx,xy,xz,yy,yz,zz) (A.1 step 3;
    scene/gaussian_model.py:27-31 + utils/general_utils.py:80-120)."""
    R = quat_to_rotmat(rots)
    L = R * (mod * scales)[:, None, :]          # R @ diag(s)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


def _xf43(p: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """Row-vector convention: matrices arrive transposed and flat (A.0):
    x' = m[0]x + m[4]y + m[8]z + m[12]  ==  (p,1) @ M[:, :3]."""
    return p @ m[:3, :3] + m[3, :3]


def preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
               cov3D_precomp, viewmatrix, projmatrix, campos, *, W: int, H: int,
               tanfovx: float, tanfovy: float, sh_degree: int, scale_modifier: float = 1.0
               ) -> Dict[str, torch.Tensor]:
    """Per-Gaussian forward stage (A.1 steps 1-8), vectorised and differentiable."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V = viewmatrix.to(dt).reshape(4, 4)
    PV = projmatrix.to(dt).reshape(4, 4)
    p_view = _xf43(means3D, V)
    in_front = p_view[:, 2] > NEAR_Z

    hom = means3D @ PV[:3, :] + PV[3, :]
    p_w = 1.0 / (hom[:, 3] + W_EPS)
    ndc = hom[:, :2] * p_w[:, None]
    if means2D is not None:                      # gradient carrier: dL/d(ndc) lands here
        ndc = ndc + means2D[:, :2]

    cov3D = cov3D_precomp if cov3D_precomp is not None else cov3d_from_scale_rot(scales, scale_modifier, rotations)

    # EWA cov2D (A.1 step 4)
    tz = p_view[:, 2]
    limx, limy = FOV_CLAMP * tanfovx, FOV_CLAMP * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    in_x = ~((txtz < -limx) | (txtz > limx))
    in_y = ~((tytz < -limy) | (tytz > limy))
    # upstream's backward treats the clamped value as a constant (x_grad_mul / y_grad_mul)
    tx = torch.where(in_x, txtz * tz, (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(in_y, tytz * tz, (tytz.clamp(-limy, limy) * tz).detach())
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz),
                     zero, fy / tz, -(fy * ty) / (tz * tz)], dim=-1).view(P, 2, 3)
    Wrot = V[:3, :3].T                            # world->view rotation, column-vector form
    Sig = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2],
                       cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                       cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], dim=-1).view(P, 3, 3)
    M = J @ Wrot
    cov2 = M @ Sig @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + LOWPASS
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + LOWPASS
    det = a * c - b * b
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=-1)
    mid = 0.5 * (a + c)
    root = torch.sqrt(torch.clamp(mid * mid - det, min=LAMBDA_FLOOR))
    lam = torch.maximum(mid + root, mid - root)
    radius = torch.ceil(RADIUS_SIGMA * torch.sqrt(lam)).detach()

    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def _rect(pv, r, g):
        lo = torch.clamp(torch.trunc((pv - r) / TILE), 0, g)
        hi = torch.clamp(torch.trunc((pv + r + (TILE - 1)) / TILE), 0, g)
        return lo.to(torch.int64), hi.to(torch.int64)

    with torch.no_grad():
        x0, x1 = _rect(px, radius, gx)
        y0, y1 = _rect(py, radius, gy)
        area = (x1 - x0) * (y1 - y0)
        valid = in_front & det_ok & (area > 0)

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_rgb(sh_degree, shs, d) + 0.5, 0.0)

    return dict(valid=valid, px=px, py=py, depth=tz, conic=conic, opacity=opacities.reshape(-1),
                rgb=rgb, radius=torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32),
                x0=x0, x1=x1, y0=y0, y1=y1, tiles=torch.where(valid, area, torch.zeros_like(area)))


def _depth_sort_key(depth: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Order of the (tile|depth-bits) radix sort with in-order duplication (A.1 step 9):
    ascending fp32 depth bits, ties by ascending Gaussian index."""
    bits = depth.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    return bits * (1 << 32) + idx


def rasterize(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, *, viewmatrix, projmatrix, campos, bg, W: int, H: int, tanfovx: float,
              tanfovy: float, sh_degree: int = 0, scale_modifier: float = 1.0, return_aux: bool = False):
    """Full forward (A.1).  Returns (color [3,H,W], radii [P] int32, depth [1,H,W], alpha [1,H,W]).
    Differentiable w.r.t. every floating input; ``means2D`` ([P,3] zeros) receives dL/d(ndc xy),
    i.e. the pixel-space gradient scaled by (0.5 W, 0.5 H) like the upstream kernel."""
    g = preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                   viewmatrix, projmatrix, campos, W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy,
                   sh_degree=sh_degree, scale_modifier=scale_modifier)
    dt = means3D.dtype
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    bg = bg.to(dt)
    color = bg[:, None, None].expand(3, H, W).clone()
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha_img = torch.zeros(1, H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    final_T = torch.ones(H, W, dtype=dt)
    vid = torch.nonzero(g["valid"]).reshape(-1)
    num_rendered = int(g["tiles"].sum())
    tile_len = torch.zeros(gy, gx, dtype=torch.int64)
    if vid.numel():
        x0, x1, y0, y1 = (g[k][vid] for k in ("x0", "x1", "y0", "y1"))
        for ty in range(gy):
            rowm = (y0 <= ty) & (y1 > ty)
            if not bool(rowm.any()):
                continue
            for tx in range(gx):
                m = rowm & (x0 <= tx) & (x1 > tx)
                ids = vid[m]
                L = ids.numel()
                if L == 0:
                    continue
                tile_len[ty, tx] = L
                order = torch.argsort(_depth_sort_key(g["depth"][ids], ids))
                ids = ids[order]
                ys = torch.arange(ty * TILE, min((ty + 1) * TILE, H))
                xs = torch.arange(tx * TILE, min((tx + 1) * TILE, W))
                PY, PX = torch.meshgrid(ys, xs, indexing="ij")
                pxf, pyf = PX.reshape(-1).to(dt), PY.reshape(-1).to(dt)
                dx = g["px"][ids][:, None] - pxf[None, :]
                dy = g["py"][ids][:, None] - pyf[None, :]
                con = g["conic"][ids]
                power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
                a_raw = g["opacity"][ids][:, None] * torch.exp(power)
                # upstream backward ignores the 0.99 clamp: straight-through
                a = a_raw + (torch.clamp(a_raw, max=ALPHA_MAX) - a_raw).detach()
                keep = (power <= 0) & (a.detach() >= ALPHA_MIN)
                a_eff = torch.where(keep, a, torch.zeros_like(a))
                one_m = 1.0 - a_eff
                T_incl = torch.cumprod(one_m, dim=0)
                T_excl = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], dim=0)
                stop = keep & ((T_excl * one_m).detach() < T_MIN)
                dead = torch.cumsum(stop.to(torch.int32), dim=0) > 0
                live = keep & ~dead
                w = torch.where(live, a_eff * T_excl, torch.zeros_like(a_eff))
                Tf = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=0)
                C = (w[:, :, None] * g["rgb"][ids][:, None, :]).sum(0)          # [npx,3]
                D = (w * g["depth"][ids][:, None]).sum(0)
                A = w.sum(0)
                pos = torch.arange(1, L + 1)[:, None].expand(L, live.shape[1])
                last = torch.where(live, pos, torch.zeros_like(pos)).max(dim=0).values
                sl = (slice(ty * TILE, ty * TILE + ys.numel()), slice(tx * TILE, tx * TILE + xs.numel()))
                shp = (ys.numel(), xs.numel())
                color[(slice(None),) + sl] = (C + Tf[:, None] * bg[None, :]).T.reshape(3, *shp)
                depth[(0,) + sl] = D.reshape(shp)
                alpha_img[(0,) + sl] = A.reshape(shp)
                n_contrib[sl] = last.reshape(shp).to(torch.int32)
                final_T[sl] = Tf.detach().reshape(shp)
    if return_aux:
        aux = dict(geom=g, n_contrib=n_contrib, final_T=final_T, num_rendered=num_rendered, tile_len=tile_len)
        return color, g["radius"], depth, alpha_img, aux
    return color, g["radius"], depth, alpha_img
