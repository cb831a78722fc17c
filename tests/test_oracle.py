"""The two oracles against each other and against analytic known answers.

C oracle (tile-exact fp32, hand-derived backward) vs dense autograd oracle: forward images,
integer outputs and every gradient.  Known-answer tests pin each rule of SURVEY.md Appendix A."""
import math

import numpy as np
import pytest
import torch

from helpers import cam_kwargs, rel_l1, seeded_image_weights, small_scene
from ggsplat import cameras as CAM
from ggsplat import sh as SH
from oracle import torch_oracle as TO
from oracle.c_oracle import COracle

TOL = 2e-5


def _both(sc, cam, bg, use_sh=True, use_cov=False, da=True):
    kw = cam_kwargs(cam, bg)
    P = sc["means3D"].shape[0]
    leaf = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "opacities")}
    args = {}
    if use_sh:
        leaf["shs"] = sc["shs"].clone().requires_grad_(True); args["shs"] = leaf["shs"]
    else:
        leaf["colors"] = sc["colors"].clone().requires_grad_(True); args["colors_precomp"] = leaf["colors"]
    if use_cov:
        leaf["cov3D"] = sc["cov"].clone().requires_grad_(True); args["cov3D_precomp"] = leaf["cov3D"]
    else:
        leaf["scales"] = sc["scales"].clone().requires_grad_(True); leaf["rotations"] = sc["rotations"].clone().requires_grad_(True)
        args["scales"], args["rotations"] = leaf["scales"], leaf["rotations"]
    leaf["means2D"] = torch.zeros(P, 3, requires_grad=True)
    col, radii, dep, alp, aux = TO.rasterize(leaf["means3D"], leaf["means2D"], leaf["opacities"], sh_degree=sc["sh_degree"],
                                             return_aux=True, **args, **kw)
    wc, wd, wa = seeded_image_weights(cam.image_width, cam.image_height)
    loss = (col * wc).sum() + ((dep * wd).sum() + (alp * wa).sum() if da else 0)
    loss.backward()
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"] if use_sh else None,
                 colors_precomp=None if use_sh else sc["colors"], scales=None if use_cov else sc["scales"],
                 rotations=None if use_cov else sc["rotations"], cov3D_precomp=sc["cov"] if use_cov else None,
                 sh_degree=sc["sh_degree"], **kw)
    g = co.backward(wc, wd if da else None, wa if da else None)
    assert co.num_rendered == aux["num_rendered"]
    assert np.array_equal(co.radii, radii.numpy())
    assert np.array_equal(co.internals()["n_contrib"].astype(np.int64), aux["n_contrib"].numpy().astype(np.int64))
    assert rel_l1(co.color, col) < TOL and rel_l1(co.depth, dep) < TOL and rel_l1(co.alpha, alp) < TOL
    for k, t in leaf.items():
        assert rel_l1(g[k].reshape(t.shape), t.grad) < TOL, k
    return co, aux


@pytest.mark.parametrize("deg", [0, 3])
def test_c_vs_autograd_sh_path(deg):
    sc, cam = small_scene(P=500, W=80, H=64, sh_degree=deg, seed=3 + deg)
    _both(sc, cam, (0.2, 0.5, 0.7))


def test_c_vs_autograd_precomp_clamp_termination():
    sc, cam = small_scene(P=350, W=70, H=50, sh_degree=1, seed=5, scale_mul=25.0, opacity_boost=3.0, cam_index=2)
    g = torch.Generator().manual_seed(1)
    sc["cov"] = TO.cov3d_from_scale_rot(sc["scales"], 1.0, sc["rotations"]).contiguous()
    sc["colors"] = torch.rand(350, 3, generator=g)
    co, aux = _both(sc, cam, (0.9, 0.1, 0.4), use_sh=False, use_cov=True)
    assert float(aux["final_T"].min()) < 2e-4             # termination rule exercised
    assert int(aux["tile_len"].max()) > 256               # more than one LDS round on the GPU side


def test_cross_stage_identities():
    """render(shs) == render(colors_precomp = clamp_min(eval_sh + 0.5, 0)) and
    render(scales, rots) == render(cov3D_precomp = Sigma(scales, rots))  (SURVEY 8c)."""
    sc, cam = small_scene(P=400, W=64, H=48, sh_degree=2, seed=9, scale_mul=6.0)
    kw = cam_kwargs(cam, (0.1, 0.1, 0.1))
    a = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                rotations=sc["rotations"], sh_degree=2, **kw)
    dirs = torch.nn.functional.normalize(sc["means3D"] - cam.camera_center[None])
    cols = torch.clamp_min(SH.eval_sh(2, sc["shs"].transpose(1, 2), dirs) + 0.5, 0.0)
    cov = TO.cov3d_from_scale_rot(sc["scales"], 1.0, sc["rotations"])
    b = COracle(means3D=sc["means3D"], opacities=sc["opacities"], colors_precomp=cols, cov3D_precomp=cov, **kw)
    assert np.array_equal(a.radii, b.radii)
    assert rel_l1(a.color, b.color) < 1e-5 and rel_l1(a.alpha, b.alpha) < 1e-6


# ---------------- analytic known-answer tests (one Gaussian / two Gaussians) -----------------
def _front_cam(W=33, H=33, f=40.0):
    return CAM.look_at_camera((0, 0, -2.0), (0, 0, 0), width=W, height=H, fx=f, fy=f, cx=W / 2, cy=H / 2, device="cpu")


def _one(opacity, scale=0.05, color=(1.0, 0.5, 0.25), bg=(0.0, 0.0, 0.0), pos=(0.0, 0.0, 0.0), cam=None):
    cam = cam or _front_cam()
    co = COracle(means3D=torch.tensor([pos]), opacities=torch.tensor([[opacity]]), colors_precomp=torch.tensor([color]),
                 scales=torch.full((1, 3), scale), rotations=torch.tensor([[1.0, 0, 0, 0]]), **cam_kwargs(cam, bg))
    return co, cam


def test_single_gaussian_at_pixel_centre():
    co, cam = _one(0.8)
    it = co.internals()
    px, py = it["xy"][0]
    assert abs(px - 16.0) < 1e-4 and abs(py - 16.0) < 1e-4       # cx - 0.5 = 16
    # at the centre pixel d = 0: alpha = opacity, C = alpha * color, A = alpha, D = alpha * z
    assert np.allclose(co.color[:, 16, 16], 0.8 * np.array([1.0, 0.5, 0.25]), atol=1e-6)
    assert abs(co.alpha[0, 16, 16] - 0.8) < 1e-6 and abs(co.depth[0, 16, 16] - 0.8 * 2.0) < 1e-5
    # screen-space sigma^2 = (f s / z)^2 + 0.3 low-pass; one pixel off-centre:
    var = (40.0 * 0.05 / 2.0) ** 2 + 0.3
    assert abs(co.alpha[0, 16, 17] - 0.8 * math.exp(-0.5 / var)) < 1e-5
    lam = var
    assert co.radii[0] == math.ceil(3 * math.sqrt(lam))


def test_alpha_clamp_and_background():
    co, _ = _one(1.0, bg=(0.2, 0.4, 0.6))
    assert abs(co.alpha[0, 16, 16] - 0.99) < 1e-6                 # min(0.99, .)
    assert np.allclose(co.color[:, 16, 16], 0.99 * np.array([1, .5, .25]) + 0.01 * np.array([.2, .4, .6]), atol=1e-6)
    far = co.color[:, 0, 0]
    assert np.allclose(far, [0.2, 0.4, 0.6], atol=1e-6)           # untouched pixels = background


def test_skip_below_one_over_255():
    co, _ = _one(1.0 / 255.0 - 1e-5)
    assert float(np.abs(co.alpha).max()) == 0.0
    co, _ = _one(1.0 / 255.0 + 1e-5)
    assert co.alpha[0, 16, 16] > 0.0


def test_two_gaussians_depth_order_and_tie_break():
    cam = _front_cam()
    kw = cam_kwargs(cam, (0, 0, 0))
    def run(z0, z1):
        return COracle(means3D=torch.tensor([[0.0, 0, z0], [0.0, 0, z1]]), opacities=torch.tensor([[0.5], [0.5]]),
                       colors_precomp=torch.tensor([[1.0, 0, 0], [0.0, 1, 0]]), scales=torch.full((2, 3), 0.05),
                       rotations=torch.tensor([[1.0, 0, 0, 0]] * 2), **kw)
    near_red = run(0.0, 0.5)        # camera at z=-2 looking +z: z=0 is nearer
    assert np.allclose(near_red.color[:, 16, 16], [0.5, 0.25, 0.0], atol=1e-5)
    near_green = run(0.5, 0.0)
    assert np.allclose(near_green.color[:, 16, 16], [0.25, 0.5, 0.0], atol=1e-5)
    tie = run(0.0, 0.0)             # equal depth bits: ascending Gaussian index (stable sort)
    assert np.allclose(tie.color[:, 16, 16], [0.5, 0.25, 0.0], atol=1e-5)
    assert list(tie.internals()["list"][:2]) == [0, 1]


def test_early_termination_excludes_the_terminating_gaussian():
    cam = _front_cam()
    n = 6                            # (1-0.99)^2 = 1e-4 -> not < 1e-4?  fp32: T after 2 = 1e-4 (>=), third stops
    co = COracle(means3D=torch.tensor([[0.0, 0, 0.1 * i] for i in range(n)]), opacities=torch.ones(n, 1),
                 colors_precomp=torch.ones(n, 3), scales=torch.full((n, 3), 0.05),
                 rotations=torch.tensor([[1.0, 0, 0, 0]] * n), **cam_kwargs(cam, (0, 0, 0)))
    it = co.internals()
    T = it["final_T"][16, 16]
    k = int(it["n_contrib"][16, 16])
    assert k < n and T >= 1e-4 * 0.999 and T * (1 - 0.99) < 1e-4
    assert abs(co.alpha[0, 16, 16] - (1 - T)) < 1e-6              # sum alpha T == 1 - T_final exactly up to rounding


def test_near_plane_cull_and_offscreen():
    cam = _front_cam()
    co = COracle(means3D=torch.tensor([[0.0, 0, -1.85], [0.0, 0, -1.75], [50.0, 0, 0]]), opacities=torch.ones(3, 1) * 0.5,
                 colors_precomp=torch.ones(3, 3), scales=torch.full((3, 3), 0.01),
                 rotations=torch.tensor([[1.0, 0, 0, 0]] * 3), **cam_kwargs(cam, (0, 0, 0)))
    assert co.radii[0] == 0          # view z = 0.15 <= 0.2
    assert co.radii[1] > 0           # view z = 0.25
    assert co.radii[2] == 0          # projects outside every tile


def test_off_centre_principal_point_shifts_image():
    W = H = 33
    cam = CAM.look_at_camera((0, 0, -2.0), (0, 0, 0), width=W, height=H, fx=40., fy=40., cx=W / 2 + 4, cy=H / 2 - 3, device="cpu")
    co, _ = _one(0.8, cam=cam)
    px, py = co.internals()["xy"][0]
    assert abs(px - 20.0) < 1e-4 and abs(py - 13.0) < 1e-4
    assert abs(co.alpha[0, 13, 20] - 0.8) < 1e-6


def test_ragged_image_sizes():
    """W, H not multiples of 16: partial tiles on the right / bottom edges."""
    sc, cam = small_scene(P=300, W=37, H=21, sh_degree=0, seed=4, scale_mul=8.0)
    _both(sc, cam, (0.3, 0.3, 0.3), da=False)


def test_scale_modifier_gradient_follows_upstream():
    """scale_modifier != 1: every gradient equals autograd's EXCEPT dL/dscales, which upstream's computeCov3D backward
    returns without the chain-rule factor `modifier` (it differentiates w.r.t. S = modifier * scale).  The oracle and
    the HIP kernel reproduce upstream: oracle dL/dscales * modifier == autograd dL/dscales."""
    mod = 0.7
    sc, cam = small_scene(P=300, W=64, H=48, sh_degree=0, seed=23, scale_mul=5.0)
    kw = cam_kwargs(cam, (1.0, 1.0, 1.0))
    leaf = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    col, radii, dep, alp = TO.rasterize(leaf["means3D"], torch.zeros(300, 3), leaf["opacities"], shs=leaf["shs"],
                                        scales=leaf["scales"], rotations=leaf["rotations"], sh_degree=0,
                                        scale_modifier=mod, **kw)
    wc, _, _ = seeded_image_weights(cam.image_width, cam.image_height)
    (col * wc).sum().backward()
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=0, scale_modifier=mod, **kw)
    g = co.backward(wc)
    assert np.array_equal(co.radii, radii.numpy())
    for k in ("means3D", "opacities", "shs", "rotations"):
        assert rel_l1(g[k].reshape(leaf[k].shape), leaf[k].grad) < TOL, k
    assert rel_l1(g["scales"] * mod, leaf["scales"].grad) < TOL
    assert rel_l1(g["scales"], leaf["scales"].grad) > 0.1          # i.e. NOT the exact derivative
