"""Rows a13 / a14 of the scope table: the s2 registration step and the s3 appearance step on the GPU
(HIP rasterizer + fused mesh binding + PyTorch loss) against the same step assembled from the
oracles on the CPU (host_oracle.mesh_bind + torch_oracle.rasterize + host_oracle loss, autograd).
Compared per step: loss terms, image, radii, EVERY parameter gradient (incl. mesh.v), the
densification statistics; the Adam step itself is checked on the gradients it consumed."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l1
from ggsplat import synthetic as S
from oracle import host_oracle as HO
from oracle import torch_oracle as TO

pytestmark = pytest.mark.gpu
TOL = 2e-4
W, H = 64, 64


def _scene(sh_degree, seed=0):
    v, f = S.skirt_mesh(24, 40, r_top=0.30, r_bottom=0.5, height=0.8, jitter=2e-3, seed=seed)   # 1920 faces
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=sh_degree, seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    params["_xyz"] = torch.randn(f.shape[0], 3, generator=g) * 0.3
    cams = S.rig_cameras(n_rings=2, n_az=3, radius=2.2, width=W, height=H, f=60.0, seed=seed)
    gt = torch.rand(3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.2).float()
    return v, f, params, cams, gt, mask


def _oracle_render(cam, xyz, scaling, rot, opacity, shs, sh_degree, bg):
    P = xyz.shape[0]
    m2d = torch.zeros(P, 3, requires_grad=True)
    color, radii, depth, alpha = TO.rasterize(
        xyz, m2d, opacity, shs=shs, scales=scaling, rotations=rot, viewmatrix=cam.world_view_transform.cpu(),
        projmatrix=cam.full_proj_transform.cpu(), campos=cam.camera_center.cpu(), bg=bg, W=W, H=H,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=sh_degree)
    return color, radii, m2d


def _photometric(image, gt, mask, lam):
    return HO.l1_loss(image, gt, mask) * (1.0 - lam), 1.0 - HO.ssim(image, gt, mask) * lam


def test_registration_step_matches_oracle_pipeline():
    from ggsplat.inner_step import DEFAULT_OPT, registration_step
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.3, "threshold_scale": 0.5})
    v, f, params, cams, gt, mask = _scene(sh_degree=0)
    model = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
    model.training_setup(opt, is_ff=True, optimizer="torch")
    bg = torch.tensor([0.0, 0.0, 0.0])
    names = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    for it, cam_i in enumerate((0, 4)):
        cam = cams[cam_i]
        for n in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(cam, n, getattr(cam, n).cuda())
        # ---- oracle pipeline from the CURRENT GPU parameters (no drift between the two sides) ----
        leaf = {n: getattr(model, n).detach().cpu().clone().requires_grad_(True) for n in names}
        mv = model.mesh.v.detach().cpu().clone().requires_grad_(True)
        xyz, scaling, rot = HO.mesh_bind(mv, f, params["binding"], leaf["_xyz"], leaf["_scaling"], leaf["_rotation"])
        shs = torch.cat((leaf["_features_dc"], leaf["_features_rest"]), 1)
        image, radii, m2d = _oracle_render(cam, xyz, scaling, rot, torch.sigmoid(leaf["_opacity"]), shs, 0, bg)
        vis = radii > 0
        l_img, l_ssim = _photometric(image, gt, mask, opt.lambda_dssim)
        l_xyz = F.relu(leaf["_xyz"][vis].norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz
        l_sc = F.relu(torch.exp(leaf["_scaling"][vis]) - opt.threshold_scale).norm(dim=1).mean() * opt.lambda_scale
        (l_img + l_ssim + l_xyz + l_sc).backward()
        # ---- GPU step (no optimizer step yet: compare the gradients it will consume) ----
        accum_before = model.xyz_gradient_accum.clone()
        out = registration_step(model, cam, gt.cuda(), mask.cuda(), bg.cuda(), opt=opt, optimizer_step=False)
        val = (lambda t: float(t.detach()) if torch.is_tensor(t) else float(t))
        assert abs(val(out["img"]) - val(l_img)) < 1e-5 and abs(val(out["ssim"]) - val(l_ssim)) < 1e-5
        assert abs(val(out["xyz"]) - val(l_xyz)) < 1e-6 and abs(val(out["scale"]) - val(l_sc)) < 1e-6
        assert np.array_equal(out["render_pkg"]["radii"].cpu().numpy(), radii.numpy())
        for n in names:
            gpu_g = getattr(model, n).grad
            if leaf[n].grad is None or float(leaf[n].grad.abs().sum()) == 0:
                assert gpu_g is None or float(gpu_g.abs().sum()) == 0
                continue
            assert rel_l1(gpu_g, leaf[n].grad) <= TOL, n
        assert rel_l1(model.mesh.v.grad, mv.grad) <= TOL
        assert rel_l1(out["render_pkg"]["viewspace_points"].grad, m2d.grad) <= TOL
        stat = (model.xyz_gradient_accum - accum_before).cpu()
        ref = torch.zeros_like(stat)
        ref[vis] = m2d.grad[vis, :2].norm(dim=-1, keepdim=True)
        assert rel_l1(stat, ref) <= TOL
        assert torch.equal(model.denom.cpu().squeeze(1) >= it + 1, vis | (model.denom.cpu().squeeze(1) >= it + 1))
        # ---- Adam(eps=1e-15) step on those gradients: first step moves every touched entry by exactly lr ----
        before = {n: getattr(model, n).detach().clone() for n in names}
        grads = {n: getattr(model, n).grad.clone() for n in names if getattr(model, n).grad is not None}
        model.optimizer.step()
        model.optimizer.zero_grad()
        if it == 0:
            lr = {"_xyz": opt.position_lr_init, "_opacity": opt.opacity_lr, "_scaling": opt.scaling_lr, "_rotation": opt.rotation_lr}
            for n, l in lr.items():
                moved = (getattr(model, n).detach() - before[n])
                big = grads[n].abs() > 1e-10
                assert torch.allclose(moved[big], -l * torch.sign(grads[n][big]), rtol=1e-3, atol=l * 1e-3), n


def test_appearance_step_matches_oracle_pipeline():
    """s3 form: barycentric origin, local_xyz = _xyz + net offset, shs = features + net offset, vis_mask gather."""
    from ggsplat.inner_step import DEFAULT_OPT, appearance_step
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.3, "threshold_scale": 0.5})
    v, f, params, cams, gt, mask = _scene(sh_degree=1, seed=3)
    P = f.shape[0]
    g = torch.Generator().manual_seed(9)
    bc = torch.rand(P, 3, generator=g)
    bc = bc / bc.sum(1, keepdim=True)
    model = MeshGaussianModel.from_tensors(v, f, params, sh_degree=1, device="cuda", gs_bc=bc)
    net_w = torch.randn(3, 3, generator=g) * 0.05           # a tiny stand-in "network": offsets linear in _xyz
    sh_off = torch.randn(P, 4, 3, generator=g) * 0.05
    vis_mask = torch.rand(P, generator=g) > 0.5
    cam = cams[2]
    for n in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, n, getattr(cam, n).cuda())
    bg = torch.tensor([0.2, 0.3, 0.1])
    wg = net_w.clone().cuda().requires_grad_(True)

    def net(gm, c):
        return gm._xyz @ wg, sh_off.cuda(), vis_mask.cuda()

    out = appearance_step(model, net, cam, gt.cuda(), mask.cuda(), bg.cuda(), opt=opt)
    # ---- oracle ----
    names = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    leaf = {n: params[n].clone().requires_grad_(True) for n in names}
    wc = net_w.clone().requires_grad_(True)
    mv = v.clone().requires_grad_(True)
    local = leaf["_xyz"] + leaf["_xyz"] @ wc
    xyz, scaling, rot = HO.mesh_bind(mv, f, params["binding"], local, leaf["_scaling"], leaf["_rotation"], bary=bc)
    shs = torch.cat((leaf["_features_dc"], leaf["_features_rest"]), 1) + sh_off
    opacity = torch.sigmoid(leaf["_opacity"])
    image, radii, m2d = _oracle_render(cam, xyz[vis_mask], scaling[vis_mask], rot[vis_mask], opacity[vis_mask], shs[vis_mask], 1, bg)
    l_img, l_ssim = _photometric(image, gt, mask, opt.lambda_dssim)
    l_xyz = F.relu(local.norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz
    l_sc = F.relu(torch.exp(leaf["_scaling"]) - opt.threshold_scale).norm(dim=1).mean() * opt.lambda_scale
    l_op = F.relu(opt.threshold_opacity - opacity).mean() * opt.lambda_opacity
    (l_img + l_ssim + l_xyz + l_sc + l_op).backward()
    assert abs(float(out["loss"].detach()) - float((l_img + l_ssim + l_xyz + l_sc + l_op).detach())) < 2e-5
    assert np.array_equal(out["render_pkg"]["radii"].cpu().numpy(), radii.numpy())
    for n in names:
        assert rel_l1(getattr(model, n).grad, leaf[n].grad) <= TOL, n
    assert rel_l1(model.mesh.v.grad, mv.grad) <= TOL
    assert rel_l1(wg.grad, wc.grad) <= TOL


def test_render_python_paths_and_doll_render_on_gpu():
    """render() with pipe.compute_cov3D_python / convert_SHs_python (precomputed cov3D and colours go through the
    other two input modes of the HIP forward) and the forward-only doll_render of inference.py."""
    from types import SimpleNamespace as NS
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.render import doll_render, render
    v, f, params, cams, gt, mask = _scene(sh_degree=1, seed=6)
    params["_rotation"] = torch.tensor([1.0, 0, 0, 0]).repeat(f.shape[0], 1)
    model = MeshGaussianModel.from_tensors(v, f, params, sh_degree=1, device="cuda")
    model.update_face_coor()
    cam = cams[1]
    for n in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, n, getattr(cam, n).cuda())
    bg = torch.tensor([0.1, 0.1, 0.1], device="cuda")
    base = render(cam, model, NS(debug=False, compute_cov3D_python=False, convert_SHs_python=False), bg)
    # SH evaluated in Python == SH evaluated in the kernel (cross-stage identity through the product's eval_sh)
    py_sh = render(cam, model, NS(debug=False, compute_cov3D_python=False, convert_SHs_python=True), bg)
    assert rel_l1(py_sh["render"], base["render"]) <= 1e-5 and torch.equal(py_sh["radii"], base["radii"])
    # cov3D python path runs (it uses the LOCAL rotation like the reference, so only shapes / finiteness are checked)
    py_cov = render(cam, model, NS(debug=False, compute_cov3D_python=True, convert_SHs_python=False), bg)
    assert py_cov["render"].shape == (3, H, W) and bool(torch.isfinite(py_cov["render"]).all())
    with torch.no_grad():
        doll = NS(xyz=model.get_xyz, opacity=model.get_opacity, scaling=model.get_scaling, rotation=model.get_rotation,
                  features=model.get_features, active_sh_degree=1)
        img, depth, alpha = doll_render(cam, doll, NS(debug=False), bg)
    assert torch.equal(img, base["render"]) and alpha.shape == (1, H, W) and depth.shape == (1, H, W)


@pytest.mark.parametrize("sh_degree,bary", [(0, False), (2, True)])
def test_model_fwd_bwd_views_matches_the_autograd_path(sh_degree, bary):
    """ggsplat.batch.model_fwd_bwd_views (what bench.py's step runs: C entry points, no autograd graph) gives the gradient
    bucket that autograd gives through the model's getters and the same multi-view render."""
    from ggsplat import batch
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    Wd, Hd = 160, 96
    v, f = S.skirt_mesh(24, 40, r_top=0.30, r_bottom=0.5, height=0.8, jitter=2e-3, seed=4)
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=sh_degree, seed=4)
    g = torch.Generator().manual_seed(12)
    bc = torch.rand(f.shape[0], 3, generator=g) + 0.05
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=sh_degree, device="cuda",
                                       gs_bc=(bc / bc.sum(1, keepdim=True)) if bary else None)
    cams = S.stack_cameras(S.rig_cameras(n_rings=2, n_az=3, radius=2.2, width=Wd, height=Hd, f=120.0, seed=4), device="cuda")
    bg = torch.zeros(3, device="cuda")
    dL = torch.randn(6, 3, Hd, Wd, generator=g).cuda()
    fn = lambda v0, v1, color: dL[v0:v1]
    lean = batch.model_fwd_bwd_views(m, cams, bg=bg, W=Wd, H=Hd, chunk=4, dL_dcolor_fn=fn)["flat"]
    plist = m.parameters()
    for p in plist:
        p.grad = None
    m.update_face_coor()
    xyz, scaling, rot, opacity, shs = m.get_xyz, m.get_scaling, m.get_rotation, m.get_opacity, m.get_features
    gr = batch.fwd_bwd_views(dict(means3D=xyz.detach(), scales=scaling.detach(), rotations=rot.detach(), opacities=opacity.detach(),
                                  shs=shs.detach()), cams, bg=bg, W=Wd, H=Hd, sh_degree=sh_degree, chunk=4, dL_dcolor_fn=fn)
    torch.autograd.backward([xyz, scaling, rot, opacity, shs],
                            [gr["means3D"], gr["scales"], gr["rotations"], gr["opacities"], gr["shs"]])
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in plist])
    assert lean.shape == ref.shape and float(ref.abs().sum()) > 0
    o = 0
    for p in plist:
        n = p.numel()
        if n and float(ref[o:o + n].abs().sum()) > 0:
            assert rel_l1(lean[o:o + n], ref[o:o + n]) <= 2e-6, (o, n)
        o += n


@pytest.mark.parametrize("captured", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
def test_pipelined_fwd_bwd_views_equals_the_serial_form(mode, captured):
    """fwd_bwd_views(pipeline=1 | 2): launch sets software-pipelined over a second stream (whole forward beside the previous
    backward / only the compositing beside it), eagerly and as parallel branches of a captured graph.  Same kernels, same
    order of the backward launches: the gradient bucket and the per-view dL/dmeans2D equal the serial form up to the order of
    the float atomics inside the render backward (the serial form's own run-to-run noise)."""
    from ggsplat import batch
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    Wd, Hd = 160, 96
    v, f = S.skirt_mesh(24, 40, r_top=0.30, r_bottom=0.5, height=0.8, jitter=2e-3, seed=4)
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], sh_degree=1, seed=4), sh_degree=1, device="cuda")
    cams = S.stack_cameras(S.rig_cameras(n_rings=2, n_az=5, radius=2.2, width=Wd, height=Hd, f=120.0, seed=4), device="cuda")
    bg = torch.zeros(3, device="cuda")
    dL = torch.randn(10, 3, Hd, Wd, generator=torch.Generator().manual_seed(3)).cuda()
    gt = torch.rand(10, 3, Hd, Wd, generator=torch.Generator().manual_seed(5)).cuda()
    # the loss gradient DEPENDS on the rendered image (an L2 term on top of a fixed weight image): the pipelined forms must hand
    # the backward of a launch set the colours of ITS forward, computed on the caller's stream behind that forward
    fn = lambda v0, v1, color: dL[v0:v1] + 2.0 * (color - gt[v0:v1])

    def run(pipeline):
        return batch.model_fwd_bwd_views(m, cams, bg=bg, W=Wd, H=Hd, chunk=3, dL_dcolor_fn=fn, pipeline=pipeline, want_means2D=True)
    serial = run(0)                        # (also the eager priming: learns the binning capacity of the launch-set shapes)
    if not captured:
        piped = run(mode)
        assert piped["num_rendered"] == serial["num_rendered"] > 0
        flat, m2d = piped["flat"].clone(), piped["means2D"].clone()
    else:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            piped = run(mode)
        for _ in range(3):
            piped["flat"].zero_()
            piped["means2D"].zero_()
            g.replay()
        flat, m2d = piped["flat"].clone(), piped["means2D"].clone()
    torch.cuda.synchronize()
    assert float(serial["flat"].abs().sum()) > 0 and float(serial["means2D"].abs().sum()) > 0
    assert rel_l1(flat, serial["flat"]) <= 2e-6
    assert rel_l1(m2d, serial["means2D"]) <= 2e-6
