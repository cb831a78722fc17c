"""Oracle restatements and product-side mirrors vs golden vectors produced by the reference's own
Python (tests/golden/make_golden.py imports utils/sh_utils.py, utils/graphics_utils.py,
utils/loss_utils.py from /root/reference).  These pin every stage of the path that exists in the
reference tree; the rasterizer core itself is 'parity unpinned' (see oracle/splat_oracle.c)."""
import os

import numpy as np
import pytest
import torch

from ggsplat import cameras as CAM
from ggsplat import sh as SH
from oracle import host_oracle as HO
from oracle import torch_oracle as TO

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_eval_oracle_matches_reference(deg):
    d = _load("sh.npz")
    sh_ref_layout = torch.tensor(d["sh"])                       # [64, 3, 25]
    dirs = torch.tensor(d["dirs"])
    sh_rast_layout = sh_ref_layout.transpose(1, 2).contiguous()  # [64, K, 3] (rasterizer layout)
    got = TO.eval_sh_rgb(deg, sh_rast_layout, dirs)
    assert np.allclose(got.numpy(), d[f"deg{deg}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_eval_mirror_matches_reference(deg):
    d = _load("sh.npz")
    got = SH.eval_sh(deg, torch.tensor(d["sh"]), torch.tensor(d["dirs"]))
    assert np.allclose(got.numpy(), d[f"deg{deg}"], rtol=1e-5, atol=1e-6)
    assert np.allclose(SH.RGB2SH(torch.tensor(d["rgb"])).numpy(), d["rgb2sh"], rtol=1e-6)
    assert np.allclose(SH.SH2RGB(torch.tensor(d["rgb"])).numpy(), d["sh2rgb"], rtol=1e-6)


def test_c_oracle_sh_colour_stage_matches_reference():
    """C oracle per-Gaussian colour == clamp_min(eval_sh + 0.5, 0) of the reference (the
    convert_SHs_python path of gaussian_renderer/__init__.py:80-85)."""
    from oracle.c_oracle import COracle
    from helpers import cam_kwargs, small_scene
    d = _load("sh.npz")
    sc, cam = small_scene(P=64, W=64, H=64, sh_degree=3, seed=0, scale_mul=3.0)
    shs = torch.tensor(d["sh"]).transpose(1, 2)[:, :16].contiguous()
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=shs, scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=3, **cam_kwargs(cam, (0, 0, 0)))
    it = co.internals()
    dirs = torch.nn.functional.normalize(sc["means3D"] - cam.camera_center[None])
    # the reference's own eval_sh applied to the same coefficients, via the golden-checked mirror
    ref = torch.clamp_min(SH.eval_sh(3, torch.tensor(d["sh"])[:, :, :16], dirs) + 0.5, 0.0).numpy()
    vis = co.radii > 0
    assert vis.sum() > 10
    assert np.allclose(it["rgb"][vis], ref[vis], rtol=1e-5, atol=1e-6)


def test_camera_matrices_match_reference():
    d = _load("cameras.npz")
    for i in range(4):
        fx, fy, cx, cy, w, h, scale = d[f"intr{i}"][:7]
        trans = d[f"intr{i}"][7:10]
        assert np.allclose([CAM.focal2fov(fx, w), CAM.focal2fov(fy, h)], d[f"fov{i}"][:2])
        assert np.allclose([CAM.fov2focal(d[f"fov{i}"][0], w), CAM.fov2focal(d[f"fov{i}"][1], h)], d[f"fov{i}"][2:])
        w2c = CAM.getWorld2View2(d[f"R{i}"], d[f"T{i}"], trans, float(scale))
        assert w2c.dtype == np.float32 and np.array_equal(w2c, d[f"w2c{i}"])
        proj = CAM.getProjectionMatrix(0.01, 100.0, d[f"fov{i}"][0], d[f"fov{i}"][1], fx, fy, cx, cy, w, h)
        assert np.array_equal(proj.numpy(), d[f"proj{i}"])
        cam = CAM.Camera(R=d[f"R{i}"], T=d[f"T{i}"], FoVx=d[f"fov{i}"][0], FoVy=d[f"fov{i}"][1], fx=fx, fy=fy, cx=cx,
                         cy=cy, image_width=int(w), image_height=int(h), trans=trans, scale=float(scale),
                         data_device="cpu")
        assert np.array_equal(cam.world_view_transform.numpy(), d[f"wvt{i}"])
        assert np.allclose(cam.full_proj_transform.numpy(), d[f"full{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(cam.camera_center.numpy(), d[f"center{i}"], rtol=1e-5, atol=1e-6)


def test_projection_gives_pixel_convention():
    """pixel = fx x/z + cx - 0.5 through the oracle's ndc2pix."""
    cam = CAM.look_at_camera((0, 0, -3), (0, 0, 0), width=640, height=480, fx=500., fy=510., cx=300., cy=250., device="cpu")
    p = torch.tensor([[0.2, -0.1, 0.5]])
    pv = p @ cam.world_view_transform[:3, :3] + cam.world_view_transform[3, :3]
    hom = torch.cat([p, torch.ones(1, 1)], 1) @ cam.full_proj_transform
    ndc = hom[:, :2] / hom[:, 3:4]
    px = ((ndc[0, 0] + 1) * 640 - 1) * 0.5
    py = ((ndc[0, 1] + 1) * 480 - 1) * 0.5
    assert abs(float(px) - (500. * float(pv[0, 0] / pv[0, 2]) + 300. - 0.5)) < 1e-3
    assert abs(float(py) - (510. * float(pv[0, 1] / pv[0, 2]) + 250. - 0.5)) < 1e-3


def test_face_orientation_matches_reference():
    d = _load("face_orientation.npz")
    R, s = HO.compute_face_orientation(torch.tensor(d["verts"]), torch.tensor(d["faces"]))
    assert np.allclose(R.numpy(), d["orientation"], rtol=1e-6, atol=1e-7)
    assert np.allclose(s.numpy(), d["scale"], rtol=1e-6, atol=1e-7)


def test_rotmat_to_quat_restatement_is_a_rotation_inverse():
    """roma is absent: check the restated rotmat->quat by round trip through the reference's own
    quaternion->matrix polynomial (utils/general_utils.py:91-109, restated in torch_oracle)."""
    d = _load("face_orientation.npz")
    R = torch.tensor(d["orientation"])
    q = HO.rotmat_to_unitquat_xyzw(R)
    q_wxyz = torch.cat([q[:, 3:], q[:, :3]], -1)
    assert np.allclose(TO.quat_to_rotmat(q_wxyz).numpy(), R.numpy(), atol=2e-6)
    assert np.allclose(q.norm(dim=1).numpy(), 1.0, atol=1e-6)


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_loss_matches_reference(tag):
    d = _load("loss.npz")
    a, b = torch.tensor(d["img1"]), torch.tensor(d["img2"])
    m = torch.tensor(d["mask"]) if tag == "mask" else None
    x = a.clone().requires_grad_(True)
    l1 = HO.l1_loss(x, b, m)
    l1.backward()
    assert abs(l1.item() - float(d[f"l1_{tag}"])) < 1e-6
    assert np.allclose(x.grad.numpy(), d[f"l1_grad_{tag}"], atol=1e-8)
    x = a.clone().requires_grad_(True)
    s = HO.ssim(x, b, m)
    s.backward()
    assert abs(s.item() - float(d[f"ssim_{tag}"])) < 1e-5
    assert np.allclose(x.grad.numpy(), d[f"ssim_grad_{tag}"], rtol=1e-4, atol=1e-8)


def _cov3d_case(tag):
    d = _load("cov3d.npz")
    s, q = torch.tensor(d["scales"]), torch.tensor(d["rots"])
    return d, s, torch.nn.functional.normalize(q), {"1": 1.0, "1p3": 1.3}[tag]


@pytest.mark.parametrize("tag", ["1", "1p3"])
def test_cov3d_matches_reference_python_path(tag):
    """scale/rotation -> cov3D pinned to the fixture generated from the reference's own build_rotation /
    build_scaling_rotation / strip_symmetric (utils/general_utils.py:74-120) composed as get_covariance composes them
    (scene/gaussian_model.py:27-31).  Both restatements are checked: the torch oracle and the C oracle's stage
    (internals()['cov3d']).  The reference normalises the quaternion inside build_rotation; the rasterizer's stage
    takes unit quaternions, so they are normalised here."""
    d, s, qn, mod = _cov3d_case(tag)
    ref = d[f"cov6_{tag}"]
    assert np.allclose(TO.quat_to_rotmat(qn).numpy(), d["R"], atol=1e-6)
    assert np.allclose(TO.cov3d_from_scale_rot(s, mod, qn).numpy(), ref, rtol=2e-5, atol=1e-9)
    from oracle.c_oracle import COracle
    from ggsplat import synthetic as S
    from helpers import cam_kwargs
    cam = S.orbit_cameras(4, width=64, img_height=48, fx=70.0, fy=70.0, cx=31.0, cy=25.0)[0]
    P = s.shape[0]
    g = torch.Generator().manual_seed(1)
    means = torch.randn(P, 3, generator=g) * 0.3
    co = COracle(means3D=means, opacities=torch.full((P, 1), 0.5), colors_precomp=torch.rand(P, 3, generator=g),
                 scales=s, rotations=qn, scale_modifier=mod, **cam_kwargs(cam, [0, 0, 0]))
    assert np.allclose(co.internals()["cov3d"], ref, rtol=2e-5, atol=1e-9)
    # ... and the API identity the reference guarantees: render(scales, rots) == render(cov3D_precomp = that Sigma)
    co2 = COracle(means3D=means, opacities=torch.full((P, 1), 0.5), colors_precomp=torch.rand(P, 3, generator=torch.Generator().manual_seed(1)),
                  cov3D_precomp=torch.tensor(ref), **cam_kwargs(cam, [0, 0, 0]))
    assert np.array_equal(co.radii, co2.radii)


@pytest.mark.parametrize("name", ["s2_xyz", "delayed", "off"])
def test_lr_schedule_matches_reference(name):
    """get_expon_lr_func (utils/general_utils.py:39-74) -- the position learning-rate schedule of training_setup."""
    from ggsplat.schedule import get_expon_lr_func
    d = _load("lr_schedule.npz")
    lr_init, lr_final, delay_steps, delay_mult, max_steps = d[name + "_cfg"]
    f = get_expon_lr_func(float(lr_init), float(lr_final), int(delay_steps), float(delay_mult), int(max_steps))
    got = np.array([f(int(t)) for t in d["steps"]])
    assert np.allclose(got, d[name], rtol=1e-12, atol=0.0)


def test_stylegan_oracle_matches_reference_native_paths():
    """oracle/stylegan_oracle.py (the checker of the full-size StyleUNet-shape GPU tests) against the goldens produced
    by the reference's own upfirdn2d_native / fused_leaky_relu CPU paths."""
    from oracle import stylegan_oracle as SO
    d = _load("stylegan_ops.npz")
    x, b = torch.tensor(d["act_x"]), torch.tensor(d["act_b"])
    assert np.allclose(SO.fused_leaky_relu(x, b).numpy(), d["act_y_bias"], rtol=1e-6, atol=1e-7)
    assert np.allclose(SO.fused_leaky_relu(x, None).numpy(), d["act_y_nobias"], rtol=1e-6, atol=1e-7)
    assert np.allclose(SO.fused_leaky_relu(torch.tensor(d["act2_x"]), torch.tensor(d["act2_b"])).numpy(), d["act2_y"], rtol=1e-6, atol=1e-7)
    inp = torch.tensor(d["ufd_in"])
    for name in ("blur", "blur3", "up2", "up2haar", "down2", "down2haar", "crop", "mixed"):
        ux, uy, dx, dy, px0, px1, py0, py1 = [int(v) for v in d[f"ufd_{name}_cfg"]]
        out = SO.upfirdn2d(inp, torch.tensor(d[f"ufd_{name}_k"]), (ux, uy), (dx, dy), (px0, px1, py0, py1))
        assert out.shape == d[f"ufd_{name}_out"].shape, name
        assert np.allclose(out.numpy(), d[f"ufd_{name}_out"], rtol=1e-5, atol=1e-6), name


# ---- host mirrors of the a13 / a1 rows, pinned to the reference's own functions (tests/golden/make_golden.py imports
# ---- scene/mesh_gaussian_model.py, scene/gaussian_model.py and gaussian_renderer/__init__.py) ------------------------------
@pytest.mark.parametrize("is_ff", [True, False])
def test_training_setup_matches_reference(is_ff):
    """MeshGaussianModel.training_setup (scene/mesh_gaussian_model.py:350-379): parameter groups, their order, learning
    rates, Adam hyper-parameters, the xyz schedule and the statistics buffers, for the first frame and for later frames."""
    from types import SimpleNamespace
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    t, d = _load("training_setup.npz"), _load("densify.npz")
    opt = SimpleNamespace(**{str(k): float(v) for k, v in zip(t["opt_keys"], t["opt_vals"])})
    opt.position_lr_max_steps = int(opt.position_lr_max_steps)
    params = {k: torch.tensor(d["p" + k]) for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}
    params["binding"] = torch.arange(d["faces"].shape[0])
    m = MeshGaussianModel.from_tensors(torch.tensor(d["verts"]), torch.tensor(d["faces"]), params, sh_degree=1, device="cpu")
    m.training_setup(opt, is_ff=is_ff, optimizer="torch")
    tag = "ff" if is_ff else "mesh"
    groups = m.optimizer.param_groups
    assert isinstance(m.optimizer, torch.optim.Adam)
    assert [g["name"] for g in groups] == [str(n) for n in t[f"{tag}_names"]]
    assert np.array_equal(np.array([g["lr"] for g in groups]), t[f"{tag}_lr"])
    assert np.array_equal(np.array([g["eps"] for g in groups]), t[f"{tag}_eps"])
    assert np.array_equal(np.array([g["betas"] for g in groups]), t[f"{tag}_betas"])
    assert [g["params"][0].numel() for g in groups] == list(t[f"{tag}_numel"])
    assert np.array_equal(np.array([m.xyz_scheduler_args(i) for i in (0, 1, 100, 7000, 30000)]), t[f"{tag}_xyz_lr_at"])
    assert m.percent_dense == float(t[f"{tag}_percent_dense"])
    assert [*m.xyz_gradient_accum.shape, *m.denom.shape] == list(t[f"{tag}_stats_shapes"])


def test_quaternion_product_is_pinned_to_the_reference_face_frame():
    """get_rotation (scene/mesh_gaussian_model.py:117-122) composes the face quaternion with the local one through roma, which
    the authoring image lacks -- but the face FRAME is pinned (face_orientation.npz: the reference's compute_face_orientation)
    and the reference's own get_xyz (:124-128) fixes the composition order as face_orien_mat[binding] @ local.  So, sign-free:
    R(get_rotation[i]) == face_orien_mat[binding[i]] @ R(normalize(_rotation[i])).  Host oracle here (rotmat -> quaternion in
    all four branches, xyzw -> wxyz, Hamilton product, normalisation); the HIP binding kernel in tests/test_gpu_mesh_bind.py."""
    from helpers import quat_wxyz_to_rotmat
    d = _load("face_orientation.npz")
    v, f, R_ref = torch.tensor(d["verts"]), torch.tensor(d["faces"]), torch.tensor(d["orientation"]).double()
    g = torch.Generator().manual_seed(17)
    P = 8 * f.shape[0]
    binding = torch.randint(0, f.shape[0], (P,), generator=g)
    raw = torch.randn(P, 4, generator=g) * torch.rand(P, 1, generator=g) * 3          # un-normalised, as the parameter is
    _, _, rot = HO.mesh_bind(v, f, binding, torch.zeros(P, 3), torch.zeros(P, 3), raw)
    assert float((rot.norm(dim=1) - 1).abs().max()) < 1e-6
    want = R_ref[binding] @ quat_wxyz_to_rotmat(raw)
    assert float((quat_wxyz_to_rotmat(rot) - want).abs().max()) <= 1e-6
    # every branch of the rotmat -> quaternion construction occurs in the fixture
    dec = torch.cat([R_ref.diagonal(dim1=1, dim2=2), R_ref.diagonal(dim1=1, dim2=2).sum(1, keepdim=True)], 1)
    assert set(dec.argmax(1).tolist()) == {0, 1, 2, 3}
    # ... and the composition order matters on this fixture (the other order is off by O(1))
    swapped = quat_wxyz_to_rotmat(raw) @ R_ref[binding]
    assert float((quat_wxyz_to_rotmat(rot) - swapped).abs().max()) > 0.1


_RENDER_SCENARIOS = {
    "default": (dict(), dict(debug=False, compute_cov3D_python=False, convert_SHs_python=False), False, False),
    "s3": (dict(vis_mask=True), dict(debug=False, compute_cov3D_python=False, convert_SHs_python=False), True, True),
    "python": (dict(scaling_modifier=0.5), dict(debug=True, compute_cov3D_python=True, convert_SHs_python=True), False, False),
    "override": (dict(override_color=True), dict(debug=False, compute_cov3D_python=False, convert_SHs_python=False), False, False),
    "override_masked": (dict(override_color=True, vis_mask=True),
                        dict(debug=False, compute_cov3D_python=False, convert_SHs_python=False), False, False),
}


@pytest.mark.parametrize("name", sorted(_RENDER_SCENARIOS))
def test_render_hands_the_rasterizer_what_the_reference_hands_it(name, monkeypatch):
    """ggsplat.render.render against a RECORDING of the reference's render() (gaussian_renderer/__init__.py:21-122) run with a
    recording rasterizer in the extension's place: the settings, which arguments are None, and every tensor argument, for
    the default path, the s3 selection (pc.shs, get_final_xyz, vis_mask), the python SH / cov3D paths with a scaling
    modifier, and override_color with and without a mask."""
    from types import SimpleNamespace as NS
    from ggsplat import render as RM
    r = _load("render_args.npz")
    calls = []

    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings):
        calls.append((settings, dict(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                     opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)))
        n, H, W = means3D.shape[0], settings.image_height, settings.image_width
        return torch.zeros(3, H, W), torch.arange(n, dtype=torch.int32) % 3, torch.zeros(1, H, W), torch.zeros(1, H, W)
    monkeypatch.setattr(RM, "rasterize_gaussians", rasterize)
    fix = {k[4:]: torch.tensor(r[k]) for k in r.files if k.startswith("fix_")}
    P = fix["_xyz"].shape[0]
    kw, pipe, with_shs, with_local = _RENDER_SCENARIOS[name]
    pc = NS(_xyz=fix["_xyz"], active_sh_degree=1, max_sh_degree=1, get_xyz=fix["get_xyz"], get_opacity=fix["get_opacity"],
            get_scaling=fix["get_scaling"], get_rotation=fix["get_rotation"], get_features=fix["get_features"],
            get_covariance=lambda mod: torch.full((P, 6), float(mod)))
    if with_shs:
        pc.shs = fix["shs"]
    if with_local:
        pc.local_xyz, pc.get_final_xyz = fix["local_xyz"], fix["get_final_xyz"]
    cam = NS(FoVx=float(r["cam_fov"][0]), FoVy=float(r["cam_fov"][1]), image_height=int(r["cam_size"][0]),
             image_width=int(r["cam_size"][1]), world_view_transform=torch.tensor(r["cam_view"]),
             full_proj_transform=torch.tensor(r["cam_proj"]), camera_center=torch.tensor(r["cam_center"]))
    kw = dict(kw)
    if kw.get("vis_mask"):
        kw["vis_mask"] = torch.tensor(r["vis_mask"])
    if kw.get("override_color"):
        kw["override_color"] = fix["override"]
    bg = torch.tensor(r["bg"])
    out = RM.render(cam, pc, NS(**pipe), bg, **kw)
    rs, args = calls[-1]
    got = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree,
                    float(rs.prefiltered), float(rs.debug)], dtype=np.float64)
    assert np.array_equal(got, r[f"{name}_settings"])
    assert rs.bg is bg and rs.viewmatrix is cam.world_view_transform and rs.projmatrix is cam.full_proj_transform
    assert rs.campos is cam.camera_center
    assert sorted(k for k, v in args.items() if v is None) == [str(k) for k in r[f"{name}_none"]]
    for k, v in args.items():
        if v is not None:
            ref = r[f"{name}_arg_{k}"]
            assert tuple(v.shape) == ref.shape, k
            assert np.allclose(v.detach().numpy(), ref, rtol=1e-6, atol=1e-7), k
    assert bool(args["means2D"].requires_grad) == bool(r[f"{name}_means2D_requires_grad"])
    # the reference's keys + "tile_count" (the list lengths of THIS forward for the region-of-interest loss)
    assert sorted(set(out) - {"tile_count"}) == [str(k) for k in r[f"{name}_out_keys"]]
    assert np.array_equal(out["visibility_filter"].numpy(), r[f"{name}_visibility"])
    assert out["viewspace_points"].shape == (P, 3) and out["viewspace_points"].requires_grad


@pytest.mark.parametrize("name", ["doll_default", "doll_override_shs", "doll_override_color", "doll_masked"])
def test_doll_render_hands_the_rasterizer_what_the_reference_hands_it(name, monkeypatch):
    """ggsplat.render.doll_render (row a2) against a recording of the reference's doll_render (gaussian_renderer/__init__.py:
    124-221): settings, which arguments are None, every tensor argument; (image, depth, alpha) comes back."""
    from types import SimpleNamespace as NS
    from ggsplat import render as RM
    r = _load("render_args.npz")
    calls = []

    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings):
        calls.append((settings, dict(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                     opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)))
        n, H, W = means3D.shape[0], settings.image_height, settings.image_width
        return torch.zeros(3, H, W), torch.arange(n, dtype=torch.int32) % 3, torch.zeros(1, H, W), torch.zeros(1, H, W)
    monkeypatch.setattr(RM, "rasterize_gaussians", rasterize)
    fix = {k[4:]: torch.tensor(r[k]) for k in r.files if k.startswith("fix_")}
    doll = NS(xyz=fix["get_xyz"], opacity=fix["get_opacity"], scaling=fix["get_scaling"], rotation=fix["get_rotation"],
              features=fix["get_features"], active_sh_degree=1, max_sh_degree=1)
    cam = NS(FoVx=float(r["cam_fov"][0]), FoVy=float(r["cam_fov"][1]), image_height=int(r["cam_size"][0]),
             image_width=int(r["cam_size"][1]), world_view_transform=torch.tensor(r["cam_view"]),
             full_proj_transform=torch.tensor(r["cam_proj"]), camera_center=torch.tensor(r["cam_center"]))
    kw = {"doll_default": {}, "doll_override_shs": dict(override_shs=fix["shs"]),
          "doll_override_color": dict(override_color=fix["override"]),
          "doll_masked": dict(override_shs=fix["shs"], vis_mask=torch.tensor(r["vis_mask"]))}[name]
    out = RM.doll_render(cam, doll, NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False), torch.tensor(r["bg"]), **kw)
    rs, args = calls[-1]
    got = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree,
                    float(rs.prefiltered), float(rs.debug)], dtype=np.float64)
    assert np.array_equal(got, r[f"{name}_settings"])
    assert sorted(k for k, v in args.items() if v is None) == [str(k) for k in r[f"{name}_none"]]
    for k, v in args.items():
        if v is not None:
            ref = r[f"{name}_arg_{k}"]
            assert tuple(v.shape) == ref.shape and np.allclose(v.detach().numpy(), ref, rtol=1e-6, atol=1e-7), k
    assert len(out) == int(r[f"{name}_n_outputs"]) and [list(o.shape) for o in out] == r[f"{name}_out_shapes"].tolist()


# ---- the inner-loop BODIES of s2_registration.py / s3_appearance.py, executed from the reference scripts (tests/golden/make_golden.py
# ---- loop_golden) with render() / the network replaced by seeded differentiable stand-ins stored in tests/golden/loops.npz ----------
def _stub_render(d, tag, feats):
    H, W = (int(x) for x in d["HW"])
    Wc, Cv, radii = torch.tensor(d[tag + "_stub_Wc"]), torch.tensor(d[tag + "_stub_Cv"]), torch.tensor(d[tag + "_stub_radii"])

    def render(viewpoint_cam, gaussians, pipe, bg, **kw):
        z = torch.cat([f(gaussians).reshape(-1) for f in feats])
        vsp = torch.zeros_like(gaussians._xyz, requires_grad=True)
        image = torch.sigmoid(Wc @ z).view(3, H, W) * 0.9 + 0.01 * (vsp * Cv).sum()
        return {"render": image, "viewspace_points": vsp, "visibility_filter": radii > 0, "radii": radii}
    return render


def _mostly_close(a, ref, what):
    """After ONE Adam step with eps 1e-15 every element moved by +-lr: an element whose gradient is rounding noise may take the
    other sign in the two implementations -- 99.8 % of a tensor must agree, the rest by no more than two learning rates."""
    a, ref = a.detach().double(), torch.as_tensor(ref).double()
    ok = (a - ref).abs() <= 1e-7 + 1e-5 * ref.abs()
    assert float(ok.double().mean()) >= 0.998, (what, float(ok.double().mean()))
    assert float((a - ref).abs().max()) <= 0.11, what


def test_registration_step_matches_the_reference_loop_body(monkeypatch):
    """ggsplat.inner_step.registration_step (row a13) against ONE iteration of the loop body of s2_registration.py (lines 238-327 of
    the reference: update_face_coor ... optimizer.zero_grad()) executed from the script on the reference's own model class: the four
    loss terms, the screen-space gradient, every parameter and first moment after the Adam step, the densification statistics."""
    from types import SimpleNamespace as NS
    from ggsplat import inner_step as IS
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    d, t = _load("loops.npz"), _load("training_setup.npz")
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    params = {k: torch.tensor(d["p" + k]) for k in names}
    params["binding"] = torch.arange(d["faces"].shape[0])
    m = MeshGaussianModel.from_tensors(torch.tensor(d["verts"]), torch.tensor(d["faces"]), params, sh_degree=1, device="cpu")
    o = {str(k): float(v) for k, v in zip(t["opt_keys"], t["opt_vals"])}
    lam, thr_xyz, lam_xyz, thr_scale, lam_scale = (float(x) for x in d["s2_opt"])
    opt = NS(**o, lambda_dssim=lam, threshold_xyz=thr_xyz, lambda_xyz=lam_xyz, threshold_scale=thr_scale, lambda_scale=lam_scale,
             only_foreground_loss=True)
    opt.position_lr_max_steps = int(opt.position_lr_max_steps)
    m.training_setup(opt, is_ff=True, optimizer="torch")
    feats = (lambda g: g._features_dc.mean(0), lambda g: g._opacity.mean(0), lambda g: g._xyz.mean(0), lambda g: g._scaling.mean(0),
             lambda g: g._rotation.mean(0), lambda g: g.mesh.v.mean(0))
    monkeypatch.setattr(IS, "render", _stub_render(d, "s2", feats))
    out = IS.registration_step(m, NS(), torch.tensor(d["gt"]), torch.tensor(d["mask"]), torch.zeros(3), opt=opt,
                               first_frame_template=True, track_densification=True, fused_loss=False)
    for k in ("img", "ssim", "xyz", "scale"):
        assert abs(float(out[k].detach()) - float(d["s2_loss_" + k])) <= 2e-6 * max(1.0, abs(float(d["s2_loss_" + k]))), k
    assert np.allclose(out["render_pkg"]["viewspace_points"].grad.numpy(), d["s2_vsp_grad"], rtol=1e-4, atol=1e-9)
    for n in names:
        _mostly_close(getattr(m, n), d["s2" + n], n)
        st = m.optimizer.state.get(getattr(m, n), {})
        m1 = st["exp_avg"] if "exp_avg" in st else torch.zeros_like(getattr(m, n))
        assert np.allclose(m1.numpy(), d["s2" + n + "_m1"], rtol=1e-4, atol=1e-10), n
    _mostly_close(m.mesh.v, d["s2_verts"], "mesh.v")
    assert np.allclose(m.optimizer.state[m.mesh.v]["exp_avg"].numpy(), d["s2_verts_m1"], rtol=1e-4, atol=1e-10)
    assert np.array_equal(m.max_radii2D.numpy(), d["s2_max_radii2D"]) and np.array_equal(m.denom.numpy(), d["s2_denom"])
    assert np.allclose(m.xyz_gradient_accum.numpy(), d["s2_accum"], rtol=1e-4, atol=1e-10)


def test_appearance_step_matches_the_reference_loop_body(monkeypatch):
    """ggsplat.inner_step.appearance_step (row a14) against ONE iteration of the loop body of s3_appearance.py (lines 121-147: network
    call, render(vis_mask), five-term loss, backward, optimiser step) executed from the script on the reference's
    AvatarGaussianModel, the network replaced by two offset tensors as AvatarNet.forward leaves them (scene/avatar_net.py:82-87)."""
    from types import SimpleNamespace as NS
    from ggsplat import inner_step as IS
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    d = _load("loops.npz")
    names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")
    params = {k: torch.tensor(d["p" + k]) for k in names}
    params["_features_rest"] = torch.tensor(d["s3_features_rest"])
    params["binding"] = torch.arange(d["faces"].shape[0])
    m = MeshGaussianModel.from_tensors(torch.tensor(d["verts"]), torch.tensor(d["faces"]), params, sh_degree=1, device="cpu")
    xyz_off = torch.nn.Parameter(torch.tensor(d["s3_net_xyz_off"]))
    sh_off = torch.nn.Parameter(torch.tensor(d["s3_net_sh_off"]))
    vis = torch.tensor(d["s3_net_vis"])
    optim = torch.optim.Adam([{"params": [xyz_off], "lr": 1e-4}, {"params": [sh_off], "lr": 2e-3}, {"params": [m._opacity], "lr": 1e-2},
                              {"params": [m._scaling], "lr": 2e-3}, {"params": [m._features_dc], "lr": 2.5e-3}], lr=0.0, eps=1e-15)
    lam, thr_xyz, lam_xyz, thr_scale, lam_scale, thr_op, lam_op = (float(x) for x in d["s3_opt"])
    opt = NS(lambda_dssim=lam, threshold_xyz=thr_xyz, lambda_xyz=lam_xyz, threshold_scale=thr_scale, lambda_scale=lam_scale,
             threshold_opacity=thr_op, lambda_opacity=lam_op, only_foreground_loss=True)
    feats = (lambda g: g.shs.mean(0).reshape(-1), lambda g: g.get_opacity.mean(0), lambda g: g.local_xyz.mean(0), lambda g: g._scaling.mean(0))
    monkeypatch.setattr(IS, "render", _stub_render(d, "s3", feats))
    out = IS.appearance_step(m, lambda g, cam: (xyz_off, sh_off, vis), NS(), torch.tensor(d["gt"]), torch.tensor(d["mask"]),
                             torch.zeros(3), optimizer=optim, opt=opt, fused_loss=False)
    for k in ("img", "ssim", "xyz", "scale", "opacity"):
        assert abs(float(out[k].detach()) - float(d["s3_loss_" + k])) <= 2e-6 * max(1.0, abs(float(d["s3_loss_" + k]))), k
    for n, p_ in (("xyz_off", xyz_off), ("sh_off", sh_off), ("_opacity", m._opacity), ("_scaling", m._scaling), ("_features_dc", m._features_dc)):
        _mostly_close(p_, d["s3_after_" + n], n)
        assert np.allclose(optim.state[p_]["exp_avg"].numpy(), d["s3_m1_" + n], rtol=1e-4, atol=1e-10), n
