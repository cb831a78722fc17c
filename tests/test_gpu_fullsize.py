"""Full BASELINE size (100k mesh-bound Gaussians, 1920x1080, config 2): ONE view directly against the C oracle
(OpenMP on the host cores: well under a second), and -- since the oracle cannot do the whole 160-view batch in test
time -- size-independent properties:
  * partition of unity: all colours == background == c  =>  image == c everywhere  (sum alpha T + T_final = 1)
  * linearity in colour: render(a x + b y) == a render(x) + b render(y)   (bg = 0)
  * adjointness: <dL/dcolors, dc> == < w, render(dc) >  -- the colour backward is the transpose of the forward
  * directional finite differences for means / opacities / scales against the analytic gradient (sanity bound)
  * batched launch == single-view launches, bit for bit
  * radii / visible counts sane; depth >= near plane where alpha > 0
"""
import math

import pytest
import torch

from ggsplat import synthetic as S

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


@pytest.fixture(scope="module")
def scene():
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    dev = "cuda"
    v, f = S.skirt_mesh()
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=0)
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device=dev)
    m.update_face_coor()
    with torch.no_grad():
        inp = dict(means3D=m.get_xyz.clone(), scales=m.get_scaling.clone(), rotations=m.get_rotation.clone(),
                   opacities=m.get_opacity.clone())
    cams = S.rig_cameras()
    ck = S.stack_cameras([cams[0], cams[47], cams[101], cams[159]], device=dev)
    return inp, ck


def _fwd(inp, ck, colors=None, bg=(0., 0., 0.), keep=False, views=slice(None), debug=False, **over):
    from ggsplat import rasterizer as R
    a = {**inp, **over}
    dev = a["means3D"].device
    return R.forward_views(a["means3D"], a["opacities"], None, colors, a["scales"], a["rotations"], None,
                           view=ck["view"][views], proj=ck["proj"][views], campos=ck["campos"][views],
                           tanfov=ck["tanfov"][views], bg=torch.tensor(bg, device=dev), W=W, H=H, sh_degree=0,
                           keep_state=keep, debug=debug)


def test_partition_of_unity_and_sanity(scene):
    inp, ck = scene
    P = inp["means3D"].shape[0]
    c = torch.tensor([0.25, 0.5, 0.75], device="cuda")
    color, radii, depth, alpha, _ = _fwd(inp, ck, colors=c.expand(P, 3).contiguous(), bg=(0.25, 0.5, 0.75))
    assert float((color - c[None, :, None, None]).abs().max()) < 2e-4
    assert int((radii > 0).sum()) == 4 * P                      # the whole garment is in front of every camera
    assert float(alpha.min()) >= 0 and float(alpha.max()) <= 1.0 + 1e-5
    covered = alpha > 0.5
    assert 0.05 < float(covered.float().mean()) < 0.4           # garment covers ~10-15 % of a 1080p frame
    assert float((depth[covered] / alpha[covered]).min()) > 0.2


def test_colour_linearity_and_adjoint(scene):
    from ggsplat import rasterizer as R
    inp, ck = scene
    P = inp["means3D"].shape[0]
    g = torch.Generator().manual_seed(0)
    x = torch.rand(P, 3, generator=g).cuda()
    y = torch.rand(P, 3, generator=g).cuda()
    cx, _, _, _, _ = _fwd(inp, ck, colors=x)
    cy, _, _, _, _ = _fwd(inp, ck, colors=y)
    cz, _, _, _, st = _fwd(inp, ck, colors=(0.3 * x + 1.7 * y), keep=True)
    ref = 0.3 * cx + 1.7 * cy
    assert float((cz - ref).abs().sum() / ref.abs().sum()) < 1e-5
    w = torch.randn(4, 3, H, W, generator=g).cuda()
    grads = R.backward_views(st, w, want_means2D=False)
    lhs = float((grads["colors_precomp"].double() * x.double()).sum())
    rhs = float((w.double() * cx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(rhs)


@pytest.mark.parametrize("name,eps", [("means3D", 1e-3), ("opacities", 1e-2), ("scales", 1e-2)])
def test_directional_finite_difference(scene, name, eps):
    from ggsplat import rasterizer as R
    inp, ck = scene
    P = inp["means3D"].shape[0]
    g = torch.Generator().manual_seed(3)
    col = torch.rand(P, 3, generator=g).cuda()
    # smooth per-pixel weights so that the loss is well conditioned for finite differences
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    w = torch.stack([torch.sin(3 * xx + c) * torch.cos(2 * yy) for c in range(3)]).cuda().expand(1, 3, H, W).contiguous()
    one = slice(1, 2)

    def loss(**over):
        c, _, _, _, _ = _fwd(inp, ck, colors=col, views=one, **over)
        return float((c.double() * w.double()).sum())

    _, _, _, _, st = _fwd(inp, ck, colors=col, views=one, keep=True)
    grad = R.backward_views(st, w, want_means2D=False)[name]
    # direction = sign of the analytic gradient, so the directional derivative is sum |g_i| (far above the
    # fp32 rounding noise of a 2-megapixel loss)
    d = torch.sign(grad).reshape(inp[name].shape)
    if name == "means3D":
        # a rigid shift along the camera's x axis: every view-space depth moves by the same amount, so the
        # (discontinuous, ungraded) depth ORDER of the splats is untouched and the loss stays smooth
        right = ck["view"][1].reshape(4, 4)[:3, 0]
        d = right[None, :].expand(P, 3).contiguous()
    if name == "opacities":
        d = d * 0.1
    if name == "scales":
        d = d * inp["scales"]
    hi = loss(**{name: (inp[name] + eps * d).clamp(1e-4, 0.999) if name == "opacities" else inp[name] + eps * d})
    lo = loss(**{name: (inp[name] - eps * d).clamp(1e-4, 0.999) if name == "opacities" else inp[name] - eps * d})
    fd = (hi - lo) / (2 * eps)
    an = float((grad.double() * d.reshape(grad.shape).double()).sum())
    # The rendered image is only piecewise smooth (1/255 alpha cut-off, T < 1e-4 termination, straight-through
    # 0.99 clamp): like the reference's backward, the analytic gradient ignores the jumps, the finite
    # difference integrates them -- a systematic few-percent gap at this scene density.  Sanity bound only;
    # the exact gradient checks are the oracle / autograd comparisons at small size and the adjoint test above.
    assert fd * an > 0 and abs(fd - an) <= 0.12 * max(abs(an), abs(fd)), (fd, an)


def test_batch_equals_single_views_bitwise(scene):
    inp, ck = scene
    P = inp["means3D"].shape[0]
    col = torch.rand(P, 3, generator=torch.Generator().manual_seed(5)).cuda()
    color, radii, depth, alpha, _ = _fwd(inp, ck, colors=col, bg=(0.1, 0.2, 0.3))
    for v in range(4):
        c1, r1, d1, a1, _ = _fwd(inp, ck, colors=col, bg=(0.1, 0.2, 0.3), views=slice(v, v + 1))
        assert torch.equal(c1[0], color[v]) and torch.equal(r1[0], radii[v])
        assert torch.equal(d1[0], depth[v]) and torch.equal(a1[0], alpha[v])


def test_latency_mapping_equals_throughput_mapping_bitwise(scene):
    """A launch of 8 views runs one wave per tile (4 pixels per lane), a launch of one view one wave per (tile, quadrant)
    with several list entries in flight and the transmittance advanced optimistically: same arithmetic per pixel, so the
    images, final_T and the last-contributor positions must be identical bit for bit; the gradients agree to the order of
    their float atomics."""
    from ggsplat import rasterizer as R
    inp, ck = scene
    P = inp["means3D"].shape[0]
    col = torch.rand(P, 3, generator=torch.Generator().manual_seed(6)).cuda()
    rep = {k: v[2:3].expand(8, *v.shape[1:]).contiguous() for k, v in ck.items()}        # the same camera 8 times
    from helpers import poison_lds
    poison_lds()                        # (stale LDS words must not reach the outputs: the walks park their records there)
    c8, r8, d8, a8, st8 = _fwd(inp, rep, colors=col, bg=(0.1, 0.2, 0.3), keep=True)
    poison_lds()
    c1, r1, d1, a1, st1 = _fwd(inp, ck, colors=col, bg=(0.1, 0.2, 0.3), keep=True, views=slice(2, 3))
    assert torch.isfinite(c1).all() and torch.isfinite(c8).all()
    assert torch.equal(c1[0], c8[5]) and torch.equal(d1[0], d8[5]) and torch.equal(a1[0], a8[5]) and torch.equal(r1[0], r8[5])
    s1, s8 = R.img_sections(st1), R.img_sections(st8)
    assert torch.equal(s1["final_T"][0], s8["final_T"][5]) and torch.equal(s1["n_contrib"][0], s8["n_contrib"][5])
    w = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(7)).cuda()
    g1 = R.backward_views(st1, w, want_means2D=False)
    g8 = R.backward_views(st8, w.expand(8, 3, H, W).contiguous(), want_means2D=False)
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations"):
        assert float((g1[k] * 8 - g8[k]).abs().sum() / g8[k].abs().sum()) < 1e-5, k
    # debug mode (pipe.debug) arms the self-check of the per-quadrant walks: their LDS record slice is filled with NaNs in front of
    # EVERY round (csrc/ggs_render.hip poison_slots), so a stale slot that reached an output -- the walks read one pair past the
    # last entry of a round, rounds of odd, even and 64 entries all occur at this size -- would be a NaN here
    cp, rp, dp, ap, stp = _fwd(inp, ck, colors=col, bg=(0.1, 0.2, 0.3), keep=True, views=slice(2, 3), debug=True)
    assert torch.equal(cp, c1) and torch.equal(dp, d1) and torch.equal(ap, a1)
    sp = R.img_sections(stp)
    assert torch.equal(sp["final_T"], s1["final_T"]) and torch.equal(sp["n_contrib"], s1["n_contrib"])
    n = st1.num_rendered
    assert torch.equal(R.bin_sections(stp)["ids"][:n], R.bin_sections(st1)["ids"][:n])      # the narrowed quadrant masks too
    gp = R.backward_views(stp, w, want_means2D=False)
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations"):
        assert torch.isfinite(gp[k]).all() and float((gp[k] - g1[k]).abs().sum() / g1[k].abs().sum()) < 1e-5, k


def test_one_view_against_the_c_oracle(scene):
    """The C oracle (OpenMP, all host cores) does finish ONE 1080p view of config 2 in well under a second on the GPU
    box's host: image, depth, alpha, radii and every gradient of that view against the HIP path, at the north-star
    tolerance."""
    import numpy as np
    from helpers import REL_L1_TOL, rel_l1
    from oracle.c_oracle import COracle
    from ggsplat import rasterizer as R
    inp, ck = scene
    P = inp["means3D"].shape[0]
    g = torch.Generator().manual_seed(7)
    colors = torch.rand(P, 3, generator=g)
    w = torch.randn(3, H, W, generator=g)
    vi = 1
    cam = S.rig_cameras()[47]
    color, radii, depth, alpha, st = _fwd(inp, ck, colors=colors.cuda(), keep=True, views=slice(vi, vi + 1))
    gr = R.backward_views(st, w.cuda()[None], want_means2D=True)
    ci = {k: v.cpu() for k, v in inp.items()}
    co = COracle(means3D=ci["means3D"], opacities=ci["opacities"], colors_precomp=colors, scales=ci["scales"],
                 rotations=ci["rotations"], viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                 campos=cam.camera_center, bg=torch.zeros(3), W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5),
                 tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=0)
    og = co.backward(w)
    assert np.array_equal(radii[0].cpu().numpy(), co.radii)
    assert rel_l1(color[0].cpu(), co.color) <= REL_L1_TOL
    assert rel_l1(depth[0].cpu(), co.depth.reshape(H, W)) <= REL_L1_TOL
    assert rel_l1(alpha[0].cpu(), co.alpha.reshape(H, W)) <= REL_L1_TOL
    errs = {}
    for k, ok in (("means3D", "means3D"), ("opacities", "opacities"), ("colors_precomp", "colors"), ("scales", "scales"),
                  ("rotations", "rotations")):
        errs[k] = rel_l1(gr[k].cpu().reshape(og[ok].shape), og[ok])
    errs["means2D"] = rel_l1(gr["means2D"][0].cpu().reshape(og["means2D"].shape), og["means2D"])
    # (the single-view backward is the SEGMENTED one: every segment starts from the forward's {T, C} checkpoints)
    print("\n[config 2, one 1080p view, segmented backward] gradients vs the C oracle, relative L1:", {k: f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= REL_L1_TOL, (k, v)
    # ... and far inside the bar: the checkpoints cost no accuracy that could be told from the float atomics' own noise
    assert max(errs.values()) <= 2e-5, errs
    co.close()


def test_eight_view_launch_against_the_c_oracle(scene):
    """The launch shape bench.py times: EIGHT config-2 views in one call (= the one-wave-per-tile kernels; the one-view test
    above runs the per-quadrant mapping), SH degree 0 features as in s2, gradients summed over the views -- against the C
    oracle run view by view on the host and summed in float64."""
    import numpy as np
    from helpers import REL_L1_TOL, rel_l1
    from oracle.c_oracle import COracle
    from ggsplat import rasterizer as R
    inp, _ = scene
    P = inp["means3D"].shape[0]
    g = torch.Generator().manual_seed(11)
    shs = (torch.rand(P, 1, 3, generator=g) - 0.5) / 0.28209479177387814         # RGB2SH of uniform colours
    w = torch.randn(3, H, W, generator=g)
    cams = [S.rig_cameras()[i] for i in (3, 30, 41, 77, 90, 118, 133, 158)]      # all five rings, all sides
    ck = S.stack_cameras(cams, device="cuda")
    color, radii, depth, alpha, st = R.forward_views(
        inp["means3D"], inp["opacities"], shs.cuda(), None, inp["scales"], inp["rotations"], None, view=ck["view"],
        proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device="cuda"), W=W, H=H, sh_degree=0)
    assert st.prm.n_views * ((W + 15) // 16) * ((H + 15) // 16) >= 57344       # above the per-quadrant mapping's threshold
    gr = R.backward_views(st, w.cuda()[None].expand(len(cams), 3, H, W).contiguous(), want_means2D=True)
    ci = {k: v.cpu() for k, v in inp.items()}
    acc = {}
    for vi, cam in enumerate(cams):
        co = COracle(means3D=ci["means3D"], opacities=ci["opacities"], shs=shs, scales=ci["scales"], rotations=ci["rotations"],
                     viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                     bg=torch.zeros(3), W=W, H=H, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                     sh_degree=0)
        og = co.backward(w)
        assert np.array_equal(radii[vi].cpu().numpy(), co.radii), vi
        assert rel_l1(color[vi].cpu(), co.color) <= REL_L1_TOL, vi
        assert rel_l1(depth[vi].cpu(), co.depth.reshape(H, W)) <= REL_L1_TOL, vi
        assert rel_l1(alpha[vi].cpu(), co.alpha.reshape(H, W)) <= REL_L1_TOL, vi
        assert rel_l1(gr["means2D"][vi].cpu().reshape(og["means2D"].shape), og["means2D"]) <= REL_L1_TOL, vi
        for k in ("means3D", "opacities", "shs", "scales", "rotations"):
            t = torch.as_tensor(og[k]).double()
            acc[k] = t if k not in acc else acc[k] + t
        co.close()
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_l1(gr[k].cpu().reshape(acc[k].shape), acc[k].float()) <= REL_L1_TOL, k


def test_stress_config_view_against_the_c_oracle():
    """BASELINE config 5 (stress): ~500k Gaussians, 3840x2160, SH degree 3 -- one view against the C oracle."""
    import numpy as np
    from helpers import REL_L1_TOL, rel_l1
    from oracle.c_oracle import COracle
    from ggsplat import rasterizer as R
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    W4, H4 = 3840, 2160
    v, f = S.skirt_mesh(559, 448)
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=3)
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=3, device="cuda")
    m.update_face_coor()
    with torch.no_grad():
        inp = dict(means3D=m.get_xyz.clone(), scales=m.get_scaling.clone(), rotations=m.get_rotation.clone(),
                   opacities=m.get_opacity.clone(), shs=m.get_features.clone())
    cam = S.rig_cameras(n_rings=1, n_az=4, width=W4, height=H4, f=3000.0)[1]
    ck = S.stack_cameras([cam], device="cuda")
    color, radii, depth, alpha, st = R.forward_views(
        inp["means3D"], inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"], None, view=ck["view"],
        proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device="cuda"), W=W4, H=H4,
        sh_degree=3)
    g = torch.Generator().manual_seed(3)
    w = torch.randn(3, H4, W4, generator=g)
    gr = R.backward_views(st, w.cuda()[None], want_means2D=False)
    ci = {k: t.cpu() for k, t in inp.items()}
    co = COracle(means3D=ci["means3D"], opacities=ci["opacities"], shs=ci["shs"], scales=ci["scales"],
                 rotations=ci["rotations"], viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                 campos=cam.camera_center, bg=torch.zeros(3), W=W4, H=H4, tanfovx=math.tan(cam.FoVx * 0.5),
                 tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
    og = co.backward(w)
    assert np.array_equal(radii[0].cpu().numpy(), co.radii)
    assert rel_l1(color[0].cpu(), co.color) <= REL_L1_TOL
    assert rel_l1(alpha[0].cpu(), co.alpha.reshape(H4, W4)) <= REL_L1_TOL
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_l1(gr[k].cpu().reshape(og[k].shape), og[k]) <= REL_L1_TOL, k
    co.close()


def test_work_order_is_a_permutation_and_xcd_aware(scene):
    """order[] of an eight-view launch (ggs_k_order_tiles): every (view, tile) item exactly once; the non-empty items sit at
    positions r * stride in list-length class order (longest lists first); and the placement is XCD-aware -- workgroup b runs
    on XCD b % 8, each XCD has its own L2, and the tiles of one 4 x 4 block (which share their splats' records) land on the XCD
    of their region wherever the class sizes allow: most non-empty items sit at a position p with p % 8 == region(tile)."""
    import numpy as np
    from ggsplat import rasterizer as R
    inp, _ = scene
    P = inp["means3D"].shape[0]
    shs = torch.zeros(P, 1, 3, device="cuda")
    cams = [S.rig_cameras()[i] for i in (3, 30, 41, 77, 90, 118, 133, 158)]
    ck = S.stack_cameras(cams, device="cuda")
    *_, st = R.forward_views(inp["means3D"], inp["opacities"], shs, None, inp["scales"], inp["rotations"], None, view=ck["view"],
                             proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device="cuda"), W=W, H=H,
                             sh_degree=0)
    V, gx, gy = 8, (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    sec = R.bin_sections(st)
    order = sec["order"].cpu().numpy().astype(np.int64)
    cnt = sec["tile_count"].reshape(-1).cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(order), np.arange(V * T))
    ne = cnt > 0
    NE, E = int(ne.sum()), int((~ne).sum())
    stride = ((E // NE) & ~1) + 1
    pos_ne = np.arange(NE) * stride
    items = order[pos_ne]
    assert ne[items].all() and not ne[np.setdiff1d(order, items)].any()

    def bucket(L):
        lg = int(L).bit_length() - 1
        if lg >= 12: return 0
        if lg <= 4: return 13
        if lg == 5: return 12
        return 2 * (11 - lg) + (1 - ((L >> (lg - 1)) & 1))
    classes = np.array([bucket(c) for c in cnt[items]])
    assert (np.diff(classes) >= 0).all()                                   # longest lists first
    t = items % T
    region = ((t % gx) // 4 + (t // gx) // 4 * ((gx + 3) // 4)) % 8
    frac = float((pos_ne % 8 == region).mean())
    assert frac > 0.7, frac
