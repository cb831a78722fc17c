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


# ---- the reference's MODEL CLASSES and render() run on the CPU (rows a1, a13: host logic pinned by import) ---------------------
class _CpuTorch:
    """Stands in for the name `torch` inside a reference module: everything is torch's, except that factory calls drop the
    hard-coded device="cuda" (scene/gaussian_model.py, scene/mesh_gaussian_model.py, gaussian_renderer/__init__.py allocate
    that way), and `normal` returns mean + z * std with z drawn from a seeded CPU generator and RECORDED -- the test feeds the
    same z to the product, so the reference's CPU run and the product's GPU run sample the same children."""

    def __init__(self, seed=0):
        self._gen = torch.Generator().manual_seed(seed)
        self.draws = []

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def _strip(kw):
        if str(kw.get("device", "")).startswith("cuda"):
            kw.pop("device")
        return kw

    def zeros(self, *a, **kw): return torch.zeros(*a, **self._strip(kw))
    def ones(self, *a, **kw): return torch.ones(*a, **self._strip(kw))
    def zeros_like(self, *a, **kw): return torch.zeros_like(*a, **self._strip(kw))
    def ones_like(self, *a, **kw): return torch.ones_like(*a, **self._strip(kw))
    def tensor(self, *a, **kw): return torch.tensor(*a, **self._strip(kw))
    def arange(self, *a, **kw): return torch.arange(*a, **self._strip(kw))

    def normal(self, mean, std):
        z = torch.randn(std.shape, generator=self._gen)
        self.draws.append(z)
        return mean + z * std


def _import_reference_models():
    """Imports scene.gaussian_model, scene.mesh_gaussian_model and gaussian_renderer from /root/reference on a machine without
    their third-party dependencies: empty stand-in modules are registered for the names the import statements need (trimesh,
    smplx, plyfile, open3d, roma, simple_knn, munch-based utils.defaults, the rasterizer extension).  NOTHING of a stand-in is
    ever called by the functions the goldens run -- with one exception that is the point of the exercise: the rasterizer stand-in
    RECORDS the arguments render() hands to it.  `scene` is registered as a bare package (its __init__ imports the whole
    training stack) whose submodules load from the reference's own files."""
    import types

    def stub(name, **attrs):
        m = sys.modules.get(name)
        if m is None or not getattr(m, "__golden_stub__", False):
            m = types.ModuleType(name)
            m.__golden_stub__ = True
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    def never(*a, **kw):
        raise AssertionError("a stand-in for a missing third-party module was CALLED while generating goldens")
    for name in ("trimesh", "smplx", "open3d", "simple_knn"):
        stub(name)
    stub("plyfile", PlyData=never, PlyElement=never)
    stub("simple_knn._C", distCUDA2=never)
    stub("roma", rotmat_to_unitquat=never, quat_xyzw_to_wxyz=never, quat_wxyz_to_xyzw=never, quat_product=never,
         unitquat_to_rotmat=never)
    stub("utils.defaults", DEFAULTS=types.SimpleNamespace(output_root="", data_root="", aux_root="", stage1="stage1",
                                                          stage2="stage2", stage3="stage3"))
    pkg = stub("scene")
    pkg.__path__ = [os.path.join(REF, "scene")]                      # submodules come from the reference's files
    stub("scene.cameras", Camera=object)
    rec = stub("diff_gaussian_rasterization_depth_alpha")
    import importlib
    gm = importlib.import_module("scene.gaussian_model")
    mgm = importlib.import_module("scene.mesh_gaussian_model")
    return gm, mgm, rec


def _densify_fixture(seed=0):
    """2000 faces (a 40 x 25 tube), one Gaussian per face, SH degree 1: plain torch, no dependency on the product package."""
    g = torch.Generator().manual_seed(seed)
    na, nr = 40, 25
    th = torch.arange(na).float() / na * 2 * math.pi
    rows = []
    for r in range(nr + 1):
        rad = 0.35 + 0.25 * r / nr
        rows.append(torch.stack([rad * torch.cos(th), torch.full((na,), 0.9 - 1.0 * r / nr), rad * torch.sin(th)], 1))
    verts = torch.cat(rows) + torch.randn((nr + 1) * na, 3, generator=g) * 2e-3
    faces = []
    for r in range(nr):
        for a in range(na):
            i0, i1 = r * na + a, r * na + (a + 1) % na
            faces.append([i0, i1, i0 + na] if (r + a) % 2 == 0 else [i1, i1 + na, i0 + na])
    faces = torch.tensor(faces[:2000], dtype=torch.long)
    P = faces.shape[0]
    prm = {"_xyz": torch.randn(P, 3, generator=g) * 0.05,
           "_features_dc": torch.randn(P, 1, 3, generator=g) * 0.5,
           "_features_rest": torch.randn(P, 3, 3, generator=g) * 0.1,
           "_opacity": torch.randn(P, 1, generator=g) * 1.5,
           "_scaling": torch.log(torch.rand(P, 3, generator=g) * 0.6 + 0.25),
           "_rotation": torch.randn(P, 4, generator=g)}
    prm["_opacity"][::17] = -8.0                                       # below min_opacity: pruned
    grads = {k: torch.randn(v.shape, generator=g) * 1e-3 for k, v in prm.items()}
    grads["vertex"] = torch.randn(verts.shape, generator=g) * 1e-3
    stats = {"accum": torch.rand(P, 1, generator=g) * 4e-4, "denom": torch.randint(0, 3, (P, 1), generator=g).float(),
             "radii": torch.rand(P, generator=g) * 30}
    return verts, faces, prm, grads, stats


_OPT = dict(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
            position_lr_max_steps=30_000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)


def _reference_model(gm, mgm, verts, faces, prm, sh_degree=1):
    """A reference MeshGaussianModel WITHOUT its constructor (which reads a template / point cloud from disk,
    scene/mesh_gaussian_model.py:48-88): GaussianModel.__init__ + the fields the constructor would have set."""
    import types
    from torch import nn
    m = object.__new__(mgm.MeshGaussianModel)
    gm.GaussianModel.__init__(m, sh_degree)
    m.mesh = types.SimpleNamespace(v=nn.Parameter(verts.clone()), f=faces.clone())
    for k, v in prm.items():
        setattr(m, k, nn.Parameter(v.clone()))
    m.binding = torch.arange(faces.shape[0])
    m.binding_counter = torch.ones(faces.shape[0], dtype=torch.int32)
    m.max_radii2D = torch.zeros(faces.shape[0])
    m.spatial_lr_scale = 1.0
    # update_face_coor (scene/mesh_gaussian_model.py:90-95) minus its roma call (the face quaternion, which density control
    # never reads): the same two statements with the reference's own compute_face_orientation
    m.face_center = m.mesh.v[m.mesh.f].mean(1)
    m.face_orien_mat, m.face_scaling = compute_face_orientation(m.mesh.v, m.mesh.f, return_scale=True)
    return m


def model_golden():
    """densify.npz / training_setup.npz: the reference's own training_setup, add_densification_stats, densify_and_prune
    (clone, split, prune with the never-empty-a-face rule) and prune_points run on a seeded 2000-Gaussian fixture."""
    import types
    gm, mgm, _ = _import_reference_models()
    import utils.general_utils as GU
    opt = types.SimpleNamespace(**_OPT)
    rec = {}
    # ---- training_setup: group names / learning rates / Adam hyper-parameters, is_ff True and False --------------------
    verts, faces, prm, grads, stats = _densify_fixture()
    ts = {}
    for is_ff in (True, False):
        px = _CpuTorch()
        gm.torch = mgm.torch = GU.torch = px
        try:
            m = _reference_model(gm, mgm, verts, faces, prm)
            m.training_setup(opt, is_ff)
        finally:
            gm.torch = mgm.torch = GU.torch = torch
        tag = "ff" if is_ff else "mesh"
        ts[f"{tag}_names"] = np.array([g["name"] for g in m.optimizer.param_groups])
        ts[f"{tag}_lr"] = np.array([g["lr"] for g in m.optimizer.param_groups], dtype=np.float64)
        ts[f"{tag}_eps"] = np.array([g["eps"] for g in m.optimizer.param_groups], dtype=np.float64)
        ts[f"{tag}_betas"] = np.array([g["betas"] for g in m.optimizer.param_groups], dtype=np.float64)
        ts[f"{tag}_numel"] = np.array([g["params"][0].numel() for g in m.optimizer.param_groups])
        ts[f"{tag}_xyz_lr_at"] = np.array([m.xyz_scheduler_args(i) for i in (0, 1, 100, 7000, 30000)], dtype=np.float64)
        ts[f"{tag}_percent_dense"] = np.array(m.percent_dense)
        ts[f"{tag}_stats_shapes"] = np.array([*m.xyz_gradient_accum.shape, *m.denom.shape])
    ts["opt_keys"], ts["opt_vals"] = np.array(list(_OPT)), np.array(list(_OPT.values()), dtype=np.float64)
    np.savez(os.path.join(OUT, "training_setup.npz"), **ts)

    # ---- density control --------------------------------------------------------------------------------------------
    rec.update({"verts": verts.numpy(), "faces": faces.numpy(), **{"p" + k: v.numpy() for k, v in prm.items()},
                **{"g_" + k: v.numpy() for k, v in grads.items()}, **{"s_" + k: v.numpy() for k, v in stats.items()}})
    rec["hyper"] = np.array([0.0002, 0.005, 3.0], dtype=np.float64)       # max_grad, min_opacity, extent
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def snapshot(m, tag):
        out = {}
        for n in names:
            p_ = getattr(m, n)
            out[f"{tag}{n}"] = p_.detach().numpy().copy()
            st = m.optimizer.state[p_]
            out[f"{tag}{n}_m1"], out[f"{tag}{n}_m2"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
        out[f"{tag}_binding"], out[f"{tag}_counter"] = m.binding.numpy().copy(), m.binding_counter.numpy().copy()
        out[f"{tag}_accum"], out[f"{tag}_denom"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
        out[f"{tag}_radii"] = m.max_radii2D.numpy().copy()
        return out

    def prepared(px):
        m = _reference_model(gm, mgm, verts, faces, prm)
        m.training_setup(opt, True)
        for n in names:
            getattr(m, n).grad = grads[n].clone()
        m.mesh.v.grad = grads["vertex"].clone()
        m.optimizer.step()                                             # recognisable Adam moments (lr as set up: parameters move)
        m.optimizer.zero_grad(set_to_none=True)
        # the frames follow mesh.v, which the step just moved (the s2 loop calls update_face_coor every iteration)
        m.face_center = m.mesh.v[m.mesh.f].mean(1)
        m.face_orien_mat, m.face_scaling = compute_face_orientation(m.mesh.v, m.mesh.f, return_scale=True)
        return m

    for tag, max_screen in (("none", None), ("s20", 20)):
        px = _CpuTorch(seed=123)
        gm.torch = mgm.torch = GU.torch = px
        try:
            with torch.no_grad():
                m = prepared(px)
            if tag == "none":
                rec.update(snapshot(m, "pre"))                         # the state density control starts from (after one Adam step)
                rec["pre_verts"] = m.mesh.v.detach().numpy().copy()
                # add_densification_stats (scene/gaussian_model.py:410-412) on a seeded screen-space gradient / filter
                g2 = torch.Generator().manual_seed(77)
                vsp = types.SimpleNamespace(grad=torch.randn(faces.shape[0], 3, generator=g2) * 1e-3)
                filt = torch.rand(faces.shape[0], generator=g2) > 0.4
                with torch.no_grad():
                    m.add_densification_stats(vsp, filt)
                    m.add_densification_stats(vsp, filt)
                rec["ads_grad"], rec["ads_filter"] = vsp.grad.numpy(), filt.numpy()
                rec["ads_accum"], rec["ads_denom"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
            with torch.no_grad():
                m.xyz_gradient_accum, m.denom = stats["accum"].clone(), stats["denom"].clone()
                m.max_radii2D = stats["radii"].clone()
                m.densify_and_prune(0.0002, 0.005, 3.0, max_screen)
        finally:
            gm.torch = mgm.torch = GU.torch = torch
        rec.update(snapshot(m, tag))
        rec[f"{tag}_z"] = px.draws[0].numpy()                          # the standard-normal draws of the split
        assert len(px.draws) == 1
    # ---- the getters, reset_opacity, update_learning_rate, get_covariance of the reference model on the prepared state ----------
    px = _CpuTorch()
    gm.torch = mgm.torch = GU.torch = px
    try:
        with torch.no_grad():
            m = prepared(px)
            # get_xyz / get_scaling (scene/mesh_gaussian_model.py:105-128; get_rotation needs roma, which this image lacks)
            rec["get_xyz"], rec["get_scaling"] = m.get_xyz.numpy().copy(), m.get_scaling.numpy().copy()
            rec["get_opacity"], rec["get_features"] = m.get_opacity.numpy().copy(), m.get_features.numpy().copy()
            rec["face_center"], rec["face_scaling"] = m.face_center.numpy().copy(), m.face_scaling.numpy().copy()
            # the face frame get_rotation composes with (scene/mesh_gaussian_model.py:117-122 multiplies its QUATERNION through
            # roma; the same frame as a matrix is what get_xyz uses, :124-128): R(get_rotation) = face_orien_mat[binding] R(normalize(_rotation))
            rec["face_orien_mat"] = m.face_orien_mat.numpy().copy()
            rec["getters_binding"], rec["getters_rotation_raw"] = m.binding.numpy().copy(), m._rotation.detach().numpy().copy()
            # get_covariance(scaling_modifier): the LOCAL rotation with the mesh-bound scaling (scene/gaussian_model.py:118-119)
            rec["get_covariance_1p5"] = m.get_covariance(1.5).numpy().copy()
            # update_learning_rate (scene/gaussian_model.py:171-177): the "xyz" group only
            rec["lr_iters"] = np.array([1, 500, 7000, 30000])
            rec["lr_values"] = np.array([m.update_learning_rate(int(i)) for i in rec["lr_iters"]], dtype=np.float64)
            rec["lr_groups_after"] = np.array([g["lr"] for g in m.optimizer.param_groups], dtype=np.float64)
            # reset_opacity (scene/gaussian_model.py:212-215, :261-274): clamp at 0.01, zero moments of the opacity group
            m.reset_opacity()
            rec["reset_opacity"] = m._opacity.detach().numpy().copy()
            st = m.optimizer.state[m._opacity]
            rec["reset_opacity_m1"], rec["reset_opacity_m2"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
            other = m.optimizer.state[m._scaling]
            rec["reset_other_m1_absmax"] = np.array(float(other["exp_avg"].abs().max()))
    finally:
        gm.torch = mgm.torch = GU.torch = torch
    # ---- prune_points alone: ask for EVERYTHING, then for one of two Gaussians per face ---------------------------------
    px = _CpuTorch()
    gm.torch = mgm.torch = GU.torch = px
    try:
        with torch.no_grad():
            m = prepared(px)
            P = faces.shape[0]
            m.max_radii2D = torch.zeros(P)
            m.prune_points(torch.ones(P, dtype=torch.bool))
            rec["pruneall_P"] = np.array(m._xyz.shape[0])
            rec["pruneall_counter"] = m.binding_counter.numpy().copy()
            m.percent_dense = 1e9
            m.densify_and_clone(torch.ones(P, 1), 0.5, 1.0)
            rec["cloneall_P"], rec["cloneall_counter"] = np.array(m._xyz.shape[0]), m.binding_counter.numpy().copy()
            mask = torch.zeros(2 * P, dtype=torch.bool)
            mask[::3] = True                                           # both Gaussians of some faces, one of others, none of the rest
            m.prune_points(mask)
            rec["prunesome_mask"] = mask.numpy()
            rec["prunesome_binding"], rec["prunesome_counter"] = m.binding.numpy().copy(), m.binding_counter.numpy().copy()
            rec["prunesome_xyz"] = m._xyz.detach().numpy().copy()
    finally:
        gm.torch = mgm.torch = GU.torch = torch
    np.savez_compressed(os.path.join(OUT, "densify.npz"), **rec)


def avatar_golden():
    """avatar.npz (row a9): AvatarGaussianModel.get_barycentric_3d / get_xyz / get_final_xyz (scene/avatar_gaussian_model.py:
    140-159) -- the texel-bound model of stage 3, whose origin on the face is a barycentric point instead of the face centre --
    run on a seeded fixture with several Gaussians per face."""
    import importlib
    import types
    gm, mgm, _ = _import_reference_models()
    agm = importlib.import_module("scene.avatar_gaussian_model")
    verts, faces, prm, _, _ = _densify_fixture(seed=3)
    g = torch.Generator().manual_seed(33)
    Fn = faces.shape[0]
    P = 2500
    binding = torch.randint(0, Fn, (P,), generator=g)
    bc = torch.rand(P, 3, generator=g) + 0.05
    bc = bc / bc.sum(1, keepdim=True)
    local = torch.randn(P, 3, generator=g) * 0.05
    final_local = local + torch.randn(P, 3, generator=g) * 0.01
    m = object.__new__(agm.AvatarGaussianModel)
    gm.GaussianModel.__init__(m, 3)
    m.mesh = types.SimpleNamespace(v=verts.clone(), f=faces.clone())
    m.binding = binding
    m._xyz, m.local_xyz = local, final_local
    m._scaling = torch.log(torch.rand(P, 3, generator=g) * 0.6 + 0.25)
    m.gs_bc = (bc[:, 0], bc[:, 1], bc[:, 2])
    m.face_center = m.mesh.v[m.mesh.f].mean(1)
    m.face_orien_mat, m.face_scaling = compute_face_orientation(m.mesh.v, m.mesh.f, return_scale=True)
    np.savez_compressed(os.path.join(OUT, "avatar.npz"), verts=verts.numpy(), faces=faces.numpy(), binding=binding.numpy(),
                        gs_bc=bc.numpy(), xyz=local.numpy(), local_xyz=final_local.numpy(), log_scaling=m._scaling.numpy(),
                        barycentric_3d=m.get_barycentric_3d().numpy(), get_xyz=m.get_xyz.numpy(),
                        get_final_xyz=m.get_final_xyz.numpy(), get_scaling=m.get_scaling.numpy())


def _loop_body(script, start_marker, end_marker):
    """The lines of a reference SCRIPT from the first line containing `start_marker` to the first later line containing
    `end_marker`, dedented: the inner-loop bodies of s2_registration.py / s3_appearance.py are not functions, so they are executed
    from the file (here only; the text never leaves /root/reference) in a namespace of seeded stand-ins."""
    import textwrap
    lines = open(os.path.join(REF, script)).read().splitlines(keepends=True)
    i0 = next(i for i, l in enumerate(lines) if start_marker in l)
    i1 = next(i for i in range(i0, len(lines)) if end_marker in lines[i])
    return textwrap.dedent("".join(lines[i0:i1 + 1])), (i0 + 1, i1 + 1)


def _stub_render_arrays(P, H, W, n_feat, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(Wc=torch.randn(3 * H * W, n_feat, generator=g) * 0.8, Cv=torch.randn(P, 3, generator=g) * 0.05,
                radii=(torch.rand(P, generator=g) > 0.2).int() * torch.randint(1, 40, (P,), generator=g).int())


def _stub_render(arr, H, W, feats):
    """A differentiable stand-in for render() shared (as arrays in the golden file) with the tests: the image is a fixed smooth
    function of a few parameter means + a linear term in the screen-space tensor, radii / visibility are fixed."""
    def render(viewpoint_cam, gaussians, pipe, bg, **kw):
        z = torch.cat([f(gaussians).reshape(-1) for f in feats])
        vsp = torch.zeros_like(gaussians._xyz, requires_grad=True)
        image = torch.sigmoid(arr["Wc"] @ z).view(3, H, W) * 0.9 + 0.01 * (vsp * arr["Cv"]).sum()
        radii = arr["radii"]
        return {"render": image, "viewspace_points": vsp, "visibility_filter": radii > 0, "radii": radii}
    return render


def loop_golden():
    """loops.npz: ONE iteration of the s2 loop body (s2_registration.py `gaussians.update_face_coor()` ... `optimizer.zero_grad()`)
    and of the s3 loop body (s3_appearance.py `# predict appearance` ... `avatar_net.optimizer.zero_grad()`) EXECUTED from the
    reference scripts on the reference's own model classes, with render() / the network replaced by seeded differentiable
    stand-ins (stored in the file): loss terms, parameters and Adam moments after the step, densification statistics."""
    import importlib
    import types
    NS = types.SimpleNamespace
    gm, mgm, _ = _import_reference_models()
    import utils.general_utils as GU
    import torch.nn.functional as F
    rec = {}
    H, W = 24, 32
    verts, faces, prm, _, _ = _densify_fixture(seed=8)
    P = faces.shape[0]
    g = torch.Generator().manual_seed(81)
    gt, mask = torch.rand(3, H, W, generator=g), (torch.rand(1, H, W, generator=g) > 0.25).float()
    bg = torch.tensor([0.0, 0.0, 0.0])
    cam = NS(original_image=NS(cuda=lambda: gt.clone()), gt_alpha_mask=NS(cuda=lambda: mask.clone()))
    opt = NS(**_OPT, random_background=False, only_foreground_loss=True, lambda_dssim=0.2, threshold_xyz=0.02, lambda_xyz=1e-2,
             threshold_scale=0.6, lambda_scale=1.0, densify_from_iter=10 ** 9, densification_interval=100,
             opacity_reset_interval=10 ** 9, densify_grad_threshold=0.0002)
    rec.update(verts=verts.numpy(), faces=faces.numpy(), gt=gt.numpy(), mask=mask.numpy(), HW=np.array([H, W]),
               **{"p" + k: v.numpy() for k, v in prm.items()},
               s2_opt=np.array([opt.lambda_dssim, opt.threshold_xyz, opt.lambda_xyz, opt.threshold_scale, opt.lambda_scale]))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    # ---- s2 ---------------------------------------------------------------------------------------------------------------
    feats2 = (lambda m: m._features_dc.mean(0), lambda m: m._opacity.mean(0), lambda m: m._xyz.mean(0), lambda m: m._scaling.mean(0),
              lambda m: m._rotation.mean(0), lambda m: m.mesh.v.mean(0))
    arr2 = _stub_render_arrays(P, H, W, 3 + 1 + 3 + 3 + 4 + 3, seed=82)
    rec.update({"s2_stub_" + k: v.numpy() for k, v in arr2.items()})
    body, span = _loop_body("s2_registration.py", "gaussians.update_face_coor()", "gaussians.optimizer.zero_grad()")
    rec["s2_lines"] = np.array(span)
    px = _CpuTorch()
    gm.torch = mgm.torch = GU.torch = px
    try:
        m = _reference_model(gm, mgm, verts, faces, prm)
        m.training_setup(opt, True)
        m.update_face_coor = lambda: None              # (needs roma; the stand-in render does not read the face frames)
        m.mesh.get_energy_loss = lambda args, use_body=False: {}        # cloth energies: out of scope
        m.max_radii2D = torch.zeros(P)
        ns = dict(gaussians=m, viewpoint_stack=[cam], randint=lambda a, b: 0, scene=NS(getTrainCameras=lambda: [cam], cameras_extent=3.0),
                  opt=opt, background=bg, pipe=NS(debug=False), render=_stub_render(arr2, H, W, feats2), torch=torch, F=F,
                  l1_loss=l1_loss, ssim=ssim, is_first_frame=True, args=NS(is_template_seq=True), use_body=False, iter=5,
                  iterations=100, dataset=NS(white_background=False), iter_end=NS(record=lambda: None))
        exec(compile(body, "s2_registration.py[loop body]", "exec"), ns)
    finally:
        gm.torch = mgm.torch = GU.torch = torch
    for k in ("img", "ssim", "xyz", "scale"):
        rec["s2_loss_" + k] = np.array(float(ns["loss_dict"][k]))
    rec["s2_vsp_grad"] = ns["viewspace_point_tensor"].grad.numpy().copy()
    for n in names:
        p_ = getattr(m, n)
        rec["s2" + n] = p_.detach().numpy().copy()
        st_ = m.optimizer.state.get(p_, {})               # (a parameter the stand-in image does not depend on gets no gradient, no state)
        rec["s2" + n + "_m1"] = st_["exp_avg"].numpy().copy() if "exp_avg" in st_ else np.zeros_like(rec["s2" + n])
    rec["s2_verts"], rec["s2_verts_m1"] = m.mesh.v.detach().numpy().copy(), m.optimizer.state[m.mesh.v]["exp_avg"].numpy().copy()
    rec["s2_max_radii2D"], rec["s2_accum"], rec["s2_denom"] = m.max_radii2D.numpy().copy(), m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
    # ---- s3 ---------------------------------------------------------------------------------------------------------------
    agm = importlib.import_module("scene.avatar_gaussian_model")
    K = 4
    g3 = torch.Generator().manual_seed(83)
    prm3 = dict(prm)
    prm3["_features_rest"] = torch.randn(P, K - 1, 3, generator=g3) * 0.1
    net = dict(xyz_off=torch.randn(P, 3, generator=g3) * 0.01, sh_off=torch.randn(P, K, 3, generator=g3) * 0.03,
               vis=(torch.rand(P, generator=g3) > 0.4))
    args3 = NS(only_foreground_loss=True, lambda_dssim=0.2, threshold_xyz=0.02, lambda_xyz=1e-2, threshold_scale=0.6, lambda_scale=1.0,
               threshold_opacity=0.75, lambda_opacity=0.01)
    rec.update(s3_features_rest=prm3["_features_rest"].numpy(), s3_net_xyz_off=net["xyz_off"].numpy(), s3_net_sh_off=net["sh_off"].numpy(),
               s3_net_vis=net["vis"].numpy(),
               s3_opt=np.array([args3.lambda_dssim, args3.threshold_xyz, args3.lambda_xyz, args3.threshold_scale, args3.lambda_scale,
                                args3.threshold_opacity, args3.lambda_opacity]))
    feats3 = (lambda m: m.shs.mean(0).reshape(-1), lambda m: m.get_opacity.mean(0), lambda m: m.local_xyz.mean(0), lambda m: m._scaling.mean(0))
    arr3 = _stub_render_arrays(P, H, W, 3 * K + 1 + 3 + 3, seed=84)
    rec.update({"s3_stub_" + k: v.numpy() for k, v in arr3.items()})
    body3, span3 = _loop_body("s3_appearance.py", "shadow_shs, vis_mask = avatar_net(", "avatar_net.optimizer.zero_grad()")
    rec["s3_lines"] = np.array(span3)
    from torch import nn
    m3 = object.__new__(agm.AvatarGaussianModel)
    gm.GaussianModel.__init__(m3, 1)
    for k, v in prm3.items():
        setattr(m3, k, nn.Parameter(v.clone()))
    xyz_off, sh_off = nn.Parameter(net["xyz_off"].clone()), nn.Parameter(net["sh_off"].clone())

    class Net:                                          # what AvatarNet.forward leaves behind (scene/avatar_net.py:82-87), no network
        optimizer = torch.optim.Adam([{"params": [xyz_off], "lr": 1e-4}, {"params": [sh_off], "lr": 2e-3},
                                      {"params": [m3._opacity], "lr": 1e-2}, {"params": [m3._scaling], "lr": 2e-3},
                                      {"params": [m3._features_dc], "lr": 2.5e-3}], lr=0.0, eps=1e-15)

        def __call__(self, ambient, normal, camera):
            m3.local_xyz = m3._xyz + xyz_off
            m3.shs = m3.get_features + sh_off
            return sh_off, net["vis"]
    tens = NS(cuda=lambda: None)
    ns3 = dict(avatar_net=Net(), frame_data={"ambient": tens, "normal": tens}, viewpoint_cam=cam, gaussians=m3, args=args3, bg=bg,
               render=_stub_render(arr3, H, W, feats3), torch=torch, F=F, l1_loss=l1_loss, ssim=ssim,
               logger=lambda *a, **k: None, progress_bar=None, iter=3)
    # the stand-in render ignores vis_mask (the reference gathers inside render(): gaussian_renderer/__init__.py:92-100); the
    # loop body itself only passes it through
    exec(compile(body3, "s3_appearance.py[loop body]", "exec"), ns3)
    for k in ("img", "ssim", "xyz", "scale", "opacity"):
        rec["s3_loss_" + k] = np.array(float(ns3["loss_dict"][k]))
    for n, p_ in (("xyz_off", xyz_off), ("sh_off", sh_off), ("_opacity", m3._opacity), ("_scaling", m3._scaling), ("_features_dc", m3._features_dc)):
        rec["s3_after_" + n] = p_.detach().numpy().copy()
        rec["s3_m1_" + n] = Net.optimizer.state[p_]["exp_avg"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loops.npz"), **rec)


def ply_golden():
    """ply_layout.npz (row f2): what the reference's MeshGaussianModel.save_ply(path, save_local=True) hands to plyfile
    (scene/mesh_gaussian_model.py:251-283, construct_list_of_attributes scene/gaussian_model.py:179-191) -- property order and the
    per-vertex record -- and the binding.pkl it writes, with and without a `valid_faces` filter.  plyfile is not installed: its
    stand-in RECORDS the structured array (like the rasterizer stand-in records render()'s arguments); pickle is real."""
    import pickle
    import tempfile
    import types
    gm, mgm, _ = _import_reference_models()
    recorded = []
    ply = sys.modules["plyfile"]
    ply.PlyElement = types.SimpleNamespace(describe=lambda elements, name: recorded.append((name, elements.copy())) or elements)
    ply.PlyData = lambda els: types.SimpleNamespace(write=lambda path: None)
    mgm.PlyElement, mgm.PlyData = ply.PlyElement, ply.PlyData
    verts, faces, prm, _, _ = _densify_fixture(seed=9)
    g = torch.Generator().manual_seed(91)
    P = 300
    binding = torch.randint(0, 40, (P,), generator=g)
    prm = {k: v[:P].clone() for k, v in prm.items()}
    prm["_features_rest"] = torch.randn(P, 3, 3, generator=g) * 0.1
    rec = {"p" + k: v.numpy() for k, v in prm.items()}
    rec["binding"] = binding.numpy()
    for tag, valid in (("all", []), ("valid", [3, 5, 8, 13, 21, 34])):
        m = _reference_model(gm, mgm, verts, faces, prm)
        m.binding = binding.clone()
        m.mesh.valid_faces = valid
        with tempfile.TemporaryDirectory() as tmp:
            m.save_ply(os.path.join(tmp, "pc", "local_point_cloud.ply"), save_local=True)
            with open(os.path.join(tmp, "pc", "binding.pkl"), "rb") as f:
                rec[f"{tag}_binding_pkl"] = np.asarray(pickle.load(f))
        name, el = recorded[-1]
        assert name == "vertex"
        rec[f"{tag}_names"] = np.array(el.dtype.names)
        rec[f"{tag}_records"] = np.stack([el[n] for n in el.dtype.names], 1).astype(np.float32)
    rec["valid_faces"] = np.array([3, 5, 8, 13, 21, 34])
    np.savez_compressed(os.path.join(OUT, "ply_layout.npz"), **rec)


def obj_golden():
    """obj_io.npz (row f2): the OBJ dialect of utils/io_utils.py:7-60 -- the TEXT the reference's write_obj produces for a mesh with
    and without texture coordinates (as bytes), and what its read_obj parses from it."""
    import tempfile
    import types
    for name in ("plyfile",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["plyfile"], "PlyData"):
        sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    import importlib
    io_utils = importlib.import_module("utils.io_utils")
    g = np.random.default_rng(12)
    V, Fn, Vt = 37, 50, 44
    mesh = {"vertices": g.normal(size=(V, 3)).astype(np.float32) * np.float32(1.7), "uvs": g.random((Vt, 2)).astype(np.float32),
            "faces": g.integers(0, V, (Fn, 3)), "texture_faces": g.integers(0, Vt, (Fn, 3))}
    rec = {"in_" + k: v for k, v in mesh.items()}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, m in (("uv", mesh), ("plain", {k: mesh[k] for k in ("vertices", "faces")})):
            path = os.path.join(tmp, tag + ".obj")
            io_utils.write_obj(m, path)
            rec[tag + "_text"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
            back = io_utils.read_obj(path)
            for k, v in back.items():
                rec[f"{tag}_read_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "obj_io.npz"), **rec)


def render_args_golden():
    """render_args.npz: the reference's render() (gaussian_renderer/__init__.py:21-122) run with a RECORDING rasterizer in
    place of the extension: which tensors it hands over, in which mode, for the default path, the s3 selection (pc.shs,
    get_final_xyz, vis_mask), the python SH / cov3D paths with a scaling modifier, and override_color."""
    import types
    _, _, recmod = _import_reference_models()
    calls = []

    class Settings(types.SimpleNamespace):
        pass

    class Rasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            calls.append((self.rs, kw))
            n = kw["means3D"].shape[0]
            H, W = self.rs.image_height, self.rs.image_width
            return torch.zeros(3, H, W), torch.arange(n, dtype=torch.int32) % 3, torch.zeros(1, H, W), torch.zeros(1, H, W)
    recmod.GaussianRasterizationSettings = lambda **kw: Settings(**kw)
    recmod.GaussianRasterizer = Rasterizer
    import importlib
    GR = importlib.import_module("gaussian_renderer")
    g = torch.Generator().manual_seed(11)
    P, K = 6, 4
    fix = dict(_xyz=torch.zeros(P, 3), get_xyz=torch.randn(P, 3, generator=g), get_opacity=torch.rand(P, 1, generator=g),
               get_scaling=torch.rand(P, 3, generator=g), get_rotation=torch.randn(P, 4, generator=g),
               get_features=torch.randn(P, K, 3, generator=g), shs=torch.randn(P, K, 3, generator=g),
               local_xyz=torch.randn(P, 3, generator=g), get_final_xyz=torch.randn(P, 3, generator=g) + 10,
               override=torch.rand(P, 3, generator=g))
    cam = types.SimpleNamespace(FoVx=0.9, FoVy=0.7, image_height=48, image_width=64,
                                world_view_transform=torch.randn(4, 4, generator=g), full_proj_transform=torch.randn(4, 4, generator=g),
                                camera_center=torch.randn(3, generator=g))
    mask = torch.tensor([1, 0, 1, 1, 0, 0], dtype=torch.bool)
    bg = torch.tensor([0.1, 0.2, 0.3])

    def pc(with_shs=False, with_local=False):
        o = types.SimpleNamespace(_xyz=fix["_xyz"], active_sh_degree=1, max_sh_degree=1, get_xyz=fix["get_xyz"],
                                  get_opacity=fix["get_opacity"], get_scaling=fix["get_scaling"], get_rotation=fix["get_rotation"],
                                  get_features=fix["get_features"], get_covariance=lambda mod: torch.full((P, 6), float(mod)))
        if with_shs:
            o.shs = fix["shs"]
        if with_local:
            o.local_xyz, o.get_final_xyz = fix["local_xyz"], fix["get_final_xyz"]
        return o
    NS = types.SimpleNamespace
    scenarios = {
        "default": (pc(), NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False), {}),
        "s3": (pc(True, True), NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False), dict(vis_mask=mask)),
        "python": (pc(), NS(debug=True, compute_cov3D_python=True, convert_SHs_python=True), dict(scaling_modifier=0.5)),
        "override": (pc(), NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False), dict(override_color=fix["override"])),
        "override_masked": (pc(), NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False),
                            dict(override_color=fix["override"], vis_mask=mask)),
    }
    rec = {"fix_" + k: v.numpy() for k, v in fix.items()}
    rec.update(cam_fov=np.array([cam.FoVx, cam.FoVy]), cam_size=np.array([cam.image_height, cam.image_width]),
               cam_view=cam.world_view_transform.numpy(), cam_proj=cam.full_proj_transform.numpy(),
               cam_center=cam.camera_center.numpy(), vis_mask=mask.numpy(), bg=bg.numpy())
    px = _CpuTorch()
    GR.torch = px
    try:
        for name, (model, pipe, kw) in scenarios.items():
            out = GR.render(cam, model, pipe, bg, **kw)
            rs, args = calls[-1]
            rec[f"{name}_settings"] = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier,
                                                rs.sh_degree, float(rs.prefiltered), float(rs.debug)], dtype=np.float64)
            assert rs.bg is bg and rs.viewmatrix is cam.world_view_transform and rs.projmatrix is cam.full_proj_transform
            assert rs.campos is cam.camera_center
            rec[f"{name}_none"] = np.array(sorted(k for k, v in args.items() if v is None))
            for k, v in args.items():
                if v is not None:
                    rec[f"{name}_arg_{k}"] = v.detach().numpy().copy()
            rec[f"{name}_out_keys"] = np.array(sorted(out))
            rec[f"{name}_visibility"] = out["visibility_filter"].numpy()
            rec[f"{name}_means2D_requires_grad"] = np.array(bool(args["means2D"].requires_grad))
        # doll_render (gaussian_renderer/__init__.py:124-221): attribute names xyz / opacity / scaling / rotation / features,
        # override_shs, returns (image, depth, alpha)
        doll = types.SimpleNamespace(xyz=fix["get_xyz"], opacity=fix["get_opacity"], scaling=fix["get_scaling"],
                                     rotation=fix["get_rotation"], features=fix["get_features"], active_sh_degree=1, max_sh_degree=1,
                                     covariance=lambda mod: torch.full((P, 6), float(mod)))
        plain = NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        doll_scenarios = {"doll_default": {}, "doll_override_shs": dict(override_shs=fix["shs"]),
                          "doll_override_color": dict(override_color=fix["override"]),
                          "doll_masked": dict(override_shs=fix["shs"], vis_mask=mask)}
        for name, kw in doll_scenarios.items():
            out = GR.doll_render(cam, doll, plain, bg, **kw)
            rs, args = calls[-1]
            rec[f"{name}_settings"] = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier,
                                                rs.sh_degree, float(rs.prefiltered), float(rs.debug)], dtype=np.float64)
            rec[f"{name}_none"] = np.array(sorted(k for k, v in args.items() if v is None))
            for k, v in args.items():
                if v is not None:
                    rec[f"{name}_arg_{k}"] = v.detach().numpy().copy()
            rec[f"{name}_n_outputs"] = np.array(len(out))
            rec[f"{name}_out_shapes"] = np.array([list(o.shape) for o in out])
    finally:
        GR.torch = torch
    np.savez(os.path.join(OUT, "render_args.npz"), **rec)


if __name__ == "__main__":
    sh_golden(); camera_golden(); face_golden(); loss_golden(); stylegan_golden(); schedule_golden(); cov3d_golden()
    model_golden(); avatar_golden(); loop_golden(); ply_golden(); obj_golden(); render_args_golden()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
