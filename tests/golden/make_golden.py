#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own Python
(utils/sh_utils.py, utils/graphics_utils.py, utils/loss_utils.py) from /root/reference.
Runs only in the authoring container (the reference never travels to the GPU box); the .npz
files it writes are data: seeded inputs + the reference's outputs.

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from utils.sh_utils import eval_sh, RGB2SH, SH2RGB  # noqa: E402
from utils.graphics_utils import (getWorld2View2, getProjectionMatrix, focal2fov, fov2focal,  # noqa: E402
                                  compute_face_orientation)
from utils.loss_utils import l1_loss, ssim  # noqa: E402


def sh_golden():
    g = torch.Generator().manual_seed(0)
    sh = torch.randn(64, 3, 25, generator=g)                 # reference layout [..., C, K]
    d = torch.nn.functional.normalize(torch.randn(64, 3, generator=g))
    out = {f"deg{k}": eval_sh(k, sh, d).numpy() for k in range(5)}
    rgb = torch.rand(16, 3, generator=g)
    np.savez(os.path.join(OUT, "sh.npz"), sh=sh.numpy(), dirs=d.numpy(), rgb=rgb.numpy(),
             rgb2sh=RGB2SH(rgb).numpy(), sh2rgb=SH2RGB(rgb).numpy(), **out)


def camera_golden():
    rng = np.random.default_rng(1)
    rec = {}
    for i in range(4):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rng.normal(size=3) * 2
        w, h = [(640, 480), (1920, 1080), (940, 1280), (512, 512)][i]
        fx, fy = 500.0 + 300 * i, 520.0 + 280 * i
        cx, cy = w / 2 + rng.uniform(-20, 20), h / 2 + rng.uniform(-20, 20)
        trans, scale = (np.array([0.1, -0.2, 0.3]), 1.5) if i == 3 else (np.array([0.0, 0.0, 0.0]), 1.0)
        fovx, fovy = focal2fov(fx, w), focal2fov(fy, h)
        w2c = getWorld2View2(Q, T, trans, scale)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy, fx=fx, fy=fy, cx=cx, cy=cy, w=w, h=h)
        wvt = torch.tensor(w2c).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.transpose(0, 1).unsqueeze(0))).squeeze(0)     # scene/cameras.py:59-61
        center = wvt.inverse()[3, :3]
        rec[f"R{i}"], rec[f"T{i}"] = Q, T
        rec[f"intr{i}"] = np.array([fx, fy, cx, cy, w, h, scale, *trans])
        rec[f"fov{i}"] = np.array([fovx, fovy, fov2focal(fovx, w), fov2focal(fovy, h)])
        rec[f"w2c{i}"], rec[f"proj{i}"] = w2c, proj.numpy()
        rec[f"wvt{i}"], rec[f"full{i}"], rec[f"center{i}"] = wvt.numpy(), full.numpy(), center.numpy()
    np.savez(os.path.join(OUT, "cameras.npz"), **rec)


def face_golden():
    g = torch.Generator().manual_seed(2)
    v = torch.randn(40, 3, generator=g)
    f = torch.stack([torch.randperm(40, generator=g)[:3] for _ in range(32)])
    R, s = compute_face_orientation(v, f, return_scale=True)
    np.savez(os.path.join(OUT, "face_orientation.npz"), verts=v.numpy(), faces=f.numpy(), orientation=R.numpy(),
             scale=s.numpy())


def loss_golden():
    g = torch.Generator().manual_seed(3)
    a = torch.rand(3, 48, 64, generator=g)
    b = torch.rand(3, 48, 64, generator=g)
    m = (torch.rand(1, 48, 64, generator=g) > 0.3).float()
    rec = dict(img1=a.numpy(), img2=b.numpy(), mask=m.numpy())
    for tag, mask in (("nomask", None), ("mask", m)):
        x = a.clone().requires_grad_(True)
        l1 = l1_loss(x, b, mask)
        l1.backward()
        rec[f"l1_{tag}"], rec[f"l1_grad_{tag}"] = l1.item(), x.grad.numpy().copy()
        x = a.clone().requires_grad_(True)
        x1 = x + 0                                   # ssim masks img1 / img2 IN PLACE (loss_utils.py:44-46)
        s = ssim(x1, b.clone(), mask)
        s.backward()
        rec[f"ssim_{tag}"], rec[f"ssim_grad_{tag}"] = s.item(), x.grad.numpy().copy()
    np.savez(os.path.join(OUT, "loss.npz"), **rec)




def stylegan_golden():
    """fused_leaky_relu (CPU branch of scene/styleunet/fused_act.py:118-129) and upfirdn2d_native
    (scene/styleunet/upfirdn2d.py:186-227): the reference's own PyTorch paths for its two CUDA ops.  Both modules
    import the compiled extensions at import time, so empty stand-in modules are registered first (they are never
    called: only the native / CPU code paths run)."""
    import types
    for name in ("fused", "upfirdn2d"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import importlib.util

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    fa = load("ref_fused_act", os.path.join(REF, "scene/styleunet/fused_act.py"))
    up = load("ref_upfirdn2d", os.path.join(REF, "scene/styleunet/upfirdn2d.py"))
    g = torch.Generator().manual_seed(4)
    rec = {}
    x = torch.randn(2, 6, 5, 7, generator=g)
    b = torch.randn(6, generator=g)
    rec["act_x"], rec["act_b"] = x.numpy(), b.numpy()
    rec["act_y_bias"] = fa.fused_leaky_relu(x, b).numpy()
    rec["act_y_nobias"] = fa.fused_leaky_relu(x, None).numpy()
    x2 = torch.randn(3, 10, generator=g)
    b2 = torch.randn(10, generator=g)
    rec["act2_x"], rec["act2_b"], rec["act2_y"] = x2.numpy(), b2.numpy(), fa.fused_leaky_relu(x2, b2).numpy()
    k4 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k4 = k4[None, :] * k4[:, None]
    k4 = k4 / k4.sum()
    k3 = torch.randn(3, 3, generator=g)
    k2 = torch.tensor([[1.0, 1.0], [1.0, -1.0]]) / 2
    k23 = torch.randn(2, 3, generator=g)
    cases = [  # (kernel, up, down, pad(x0,x1,y0,y1))  -- the StyleUNet configurations + stress cases
        ("blur", k4, (1, 1), (1, 1), (2, 1, 2, 1)), ("blur3", k3, (1, 1), (1, 1), (1, 1, 1, 1)),
        ("up2", k4 * 4, (2, 2), (1, 1), (2, 1, 2, 1)), ("up2haar", k2, (2, 2), (1, 1), (1, 0, 1, 0)),
        ("down2", k4, (1, 1), (2, 2), (1, 1, 1, 1)), ("down2haar", k2, (1, 1), (2, 2), (0, 0, 0, 0)),
        ("crop", k3, (1, 1), (1, 1), (-1, 2, 0, -1)), ("mixed", k23, (3, 2), (2, 3), (2, 0, 1, 3))]
    inp = torch.randn(2, 3, 9, 11, generator=g)
    rec["ufd_in"] = inp.numpy()
    for name, k, u, d, p in cases:
        rec[f"ufd_{name}_k"] = k.numpy()
        rec[f"ufd_{name}_cfg"] = np.array([*u, *d, *p])
        rec[f"ufd_{name}_out"] = up.upfirdn2d_native(inp, k, u[0], u[1], d[0], d[1], *p).numpy()
    np.savez(os.path.join(OUT, "stylegan_ops.npz"), **rec)


def schedule_golden():
    """get_expon_lr_func (utils/general_utils.py:39-74).  The module imports open3d and scene.cameras at import time:
    empty stand-ins are registered for those names (nothing of them is called by the schedule)."""
    import types
    for name in ("open3d", "scene", "scene.cameras"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["scene.cameras"].Camera = object
    from utils.general_utils import get_expon_lr_func
    steps = np.array([-1, 0, 1, 10, 99, 100, 500, 1000, 7000, 15000, 29999, 30000, 45000])
    cfgs = {"s2_xyz": dict(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30_000),
            "delayed": dict(lr_init=0.01, lr_final=0.0001, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=10_000),
            "off": dict(lr_init=0.0, lr_final=0.0)}
    rec = {"steps": steps}
    for name, kw in cfgs.items():
        f = get_expon_lr_func(**kw)
        rec[name] = np.array([float(f(int(t))) for t in steps], dtype=np.float64)
        rec[name + "_cfg"] = np.array([kw.get("lr_init"), kw.get("lr_final"), kw.get("lr_delay_steps", 0),
                                       kw.get("lr_delay_mult", 1.0), kw.get("max_steps", 1000000)], dtype=np.float64)
    np.savez(os.path.join(OUT, "lr_schedule.npz"), **rec)


def cov3d_golden():
    """build_rotation / build_scaling_rotation / strip_symmetric (utils/general_utils.py:74-120) and the covariance
    built from them exactly as scene/gaussian_model.py:27-31 does (`L = build_scaling_rotation(modifier * scaling,
    rotation); strip_symmetric(L @ L.transpose(1, 2))`).  The three functions allocate with a hard-coded
    device="cuda"; the module's `torch` name is swapped for a proxy whose zeros() drops that keyword, so the
    reference's own arithmetic runs on the CPU unchanged."""
    import types
    for name in ("open3d", "scene", "scene.cameras"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["scene.cameras"].Camera = object
    import utils.general_utils as GU

    class _CpuTorch:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def zeros(*a, **kw):
            kw.pop("device", None)
            return torch.zeros(*a, **kw)
    real = GU.torch
    GU.torch = _CpuTorch()
    try:
        g = torch.Generator().manual_seed(7)
        s = torch.rand(50, 3, generator=g) * 0.2 + 0.005
        q = torch.randn(50, 4, generator=g)                       # NOT normalised: build_rotation normalises
        rec = {"scales": s.numpy(), "rots": q.numpy(), "R": GU.build_rotation(q).numpy()}
        for tag, mod in (("1", 1.0), ("1p3", 1.3)):
            L = GU.build_scaling_rotation(mod * s, q)
            rec[f"L_{tag}"] = L.numpy()
            rec[f"cov6_{tag}"] = GU.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    finally:
        GU.torch = real
    np.savez(os.path.join(OUT, "cov3d.npz"), **rec)


if __name__ == "__main__":
    sh_golden(); camera_golden(); face_golden(); loss_golden(); stylegan_golden(); schedule_golden(); cov3d_golden()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
