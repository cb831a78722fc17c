"""GPU paths that ordinary scenes never reach: per-tile lists longer than the LDS sort capacity (global
bitonic fallback, length class 0), workgroup tile windows too large for the LDS histogram (direct-atomic
fallback), binning-capacity overflow + retry, debug mode (sync + check after every kernel)."""
import math

import numpy as np
import pytest
import torch

from helpers import REL_L1_TOL, cam_kwargs, rel_l1, seeded_image_weights, small_scene
from ggsplat import synthetic as S
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


def _hip(sc, cam, bg, weights, debug=False):
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    rs = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.tensor(bg, device=dev), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        sh_degree=sc["sh_degree"], campos=cam.camera_center.to(dev), prefiltered=False, debug=debug)
    leaf = {k: sc[k].clone().to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    m2d = torch.zeros(sc["means3D"].shape[0], 3, device=dev, requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(rs)(means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"],
                                                        shs=leaf["shs"], scales=leaf["scales"], rotations=leaf["rotations"])
    (color * weights[0].to(dev)).sum().backward()
    return color.detach().cpu(), radii.cpu(), {k: v.grad.cpu() for k, v in leaf.items()}


def _check(sc, cam, bg=(0.1, 0.2, 0.3), debug=False):
    w = seeded_image_weights(cam.image_width, cam.image_height)
    color, radii, grads = _hip(sc, cam, bg, w, debug=debug)
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=sc["sh_degree"], **cam_kwargs(cam, bg))
    og = co.backward(w[0])
    assert np.array_equal(radii.numpy(), co.radii)
    assert rel_l1(color, co.color) <= REL_L1_TOL
    for k, g in grads.items():
        assert rel_l1(g.reshape(og[k].shape), og[k]) <= REL_L1_TOL, k
    return co


def test_tile_list_longer_than_lds_sort_capacity():
    """6000 faint splats piled on one spot: tile lists of ~6000 > 4096 keys -> global-memory bitonic fallback."""
    g = torch.Generator().manual_seed(0)
    P = 6000
    sc = S.random_gaussians(P, sh_degree=0, seed=1)
    sc["means3D"] = torch.randn(P, 3, generator=g) * 0.01
    sc["scales"] = torch.full((P, 3), 0.03)
    sc["opacities"] = torch.full((P, 1), 0.02)                 # faint: nothing terminates, every splat is blended
    cam = S.orbit_cameras(4, width=64, img_height=48, fx=70., fy=70., cx=31., cy=25.)[0]
    co = _check(sc, cam)
    assert int(np.diff(co.internals()["tile_start"]).max()) > 4096


def test_workgroup_tile_window_too_large_for_lds():
    """Randomly ordered splats over a 1080p frame: the 256 splats of a workgroup span > 2048 tiles, so the
    histogram / scatter take the direct-global-atomic fallback."""
    sc = S.random_gaussians(4096, sh_degree=1, seed=5)
    sc["means3D"] = sc["means3D"] * torch.tensor([2.4, 1.3, 0.3])
    sc["scales"] = sc["scales"] * 0.7
    from ggsplat.cameras import look_at_camera
    cam = look_at_camera((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), width=1920, height=1080, fx=1500., fy=1500., cx=955.,
                         cy=545., device="cpu")
    co = _check(sc, cam)
    assert (co.radii > 0).sum() > 3000


def test_binning_capacity_overflow_retries(monkeypatch):
    from ggsplat import rasterizer as R
    sc, cam = small_scene(P=900, W=96, H=64, sh_degree=0, seed=12, scale_mul=8.0)
    monkeypatch.setattr(R, "_cap_hint", {(0, 900, 96, 64, 1): 64})        # far too small: forces overflow + one retry
    _check(sc, cam)
    assert R._cap_hint[(0, 900, 96, 64, 1)] > 64


def test_debug_mode_sync_after_each_kernel():
    sc, cam = small_scene(P=500, W=80, H=48, sh_degree=2, seed=2, scale_mul=5.0)
    _check(sc, cam, debug=True)


def test_pathological_inputs_do_not_fault():
    """NaN / inf / degenerate Gaussians (the optimisation can produce them for a step): the launch must complete without a
    memory fault or a hang, `radii` must stay an integer >= 0, and a second, clean launch afterwards must be unaffected."""
    import numpy as np
    from ggsplat import rasterizer as R
    from helpers import small_scene
    sc, cam = small_scene(P=1000, W=160, H=120, sh_degree=1, seed=21, scale_mul=5.0)
    dev = "cuda"
    t = {k: v.clone().to(dev) for k, v in sc.items() if torch.is_tensor(v)}
    nan, inf = float("nan"), float("inf")
    bad = t["means3D"].clone(); scl = t["scales"].clone(); rot = t["rotations"].clone(); op = t["opacities"].clone()
    shs = t["shs"].clone()
    bad[0] = nan; bad[1, 0] = inf; bad[2, 2] = -inf; bad[3] = 1e30
    scl[4] = 0.0; scl[5] = 1e10; scl[6] = nan; scl[7, 1] = inf; scl[8] = -1.0
    rot[9] = 0.0; rot[10] = nan; rot[11] = 1e20
    op[12] = nan; op[13] = 5.0; op[14] = -1.0; op[15] = inf
    shs[16] = nan; shs[17] = inf
    from ggsplat import synthetic as S
    ck = S.stack_cameras([cam], device=dev)
    kw = dict(view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev),
              W=160, H=120, sh_degree=1)
    color, radii, depth, alpha, st = R.forward_views(bad, op, shs, None, scl, rot, None, **kw)
    g = R.backward_views(st, torch.ones(1, 3, 120, 160, device=dev))
    torch.cuda.synchronize()
    assert int(radii.min()) >= 0
    assert color.shape == (1, 3, 120, 160) and g["means3D"].shape == (1000, 3)
    # the same buffers / library state render a clean scene correctly afterwards
    c2, r2, d2, a2, _ = R.forward_views(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None, **kw)
    c3, r3, d3, a3, _ = R.forward_views(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None, **kw)
    assert torch.isfinite(c2).all() and torch.equal(c2, c3) and torch.equal(r2, r3)


@pytest.mark.parametrize("P,W,H,V,scale", [(300, 80, 48, 2, 6.0), (4000, 200, 120, 3, 3.0), (20000, 640, 360, 5, 1.5), (50, 16, 16, 2, 20.0),
                                           (1500, 1000, 40, 4, 8.0)])
def test_work_order_is_a_permutation_for_odd_shapes(P, W, H, V, scale):
    """ggs_k_order_tiles places every (view, tile) item exactly once whatever the class / region populations look like: few tiles
    (fewer items of a class than XCDs), one tile row, very long lists, almost everything empty.  The XCD-aware placement inside
    a class is a prefix-sum bijection (ggs_region_rank); a duplicate or a hole would leave a tile uncomposited."""
    import numpy as np
    from ggsplat import rasterizer as R, synthetic as S
    sc = S.random_gaussians(P, sh_degree=0, seed=P + V)
    sc["scales"] = sc["scales"] * scale
    cams = S.orbit_cameras(V, width=W, img_height=H, fx=0.9 * W, fy=0.9 * W, cx=W / 2 - 1.5, cy=H / 2 + 0.5)
    ck = S.stack_cameras(cams, device="cuda")
    dev = "cuda"
    color, radii, depth, alpha, st = R.forward_views(
        sc["means3D"].to(dev), sc["opacities"].to(dev), sc["shs"].to(dev), None, sc["scales"].to(dev), sc["rotations"].to(dev), None,
        view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    sec = R.bin_sections(st)
    order = sec["order"].cpu().numpy().astype(np.int64)
    cnt = sec["tile_count"].reshape(-1).cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(V * T))
    ne = cnt > 0
    NE, E = int(ne.sum()), int((~ne).sum())
    if NE:
        stride = ((E // NE) & ~1) + 1
        assert ne[order[np.arange(NE) * stride]].all()
    # and the launch composited every pixel: alpha + final_T == 1 up to rounding wherever something was blended
    fT = R.img_sections(st)["final_T"]
    assert float((alpha + fT - 1.0).abs().max()) < 1e-4


@pytest.mark.parametrize("n_views", [1, 8], ids=["quad_waves", "tile_waves"])
def test_backward_never_reads_the_workspace_of_empty_tiles(n_views):
    """The forward stores the per-pixel workspace (final_T, n_contrib) only where a tile has a list -- ~90 % of the tiles of a
    view are empty and their two planes were a quarter of everything the forward wrote.  Contract: nothing on the device reads
    the workspace of an empty tile.  Here it is POISONED (NaN / huge positions) between forward and backward, in both
    mappings: the gradients must come out as without the poison, and ggs_count_blends must count the same pairs."""
    import ctypes as C
    from ggsplat import _lib, rasterizer as R
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    dev = "cuda"
    W, H = 640, 360
    v, f = S.skirt_mesh(40, 60)
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], 0), 0, device=dev)
    cams = S.rig_cameras(n_rings=1, n_az=8, width=W, height=H, f=500.0)[:n_views]
    ck = S.stack_cameras(cams, device=dev)
    dL = torch.randn(n_views, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)

    def run(poison):
        with torch.no_grad():
            m.update_face_coor()
            color, radii, depth, alpha, st = R.forward_views(
                m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None, view=ck["view"],
                proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
            gx, gy = (W + 15) // 16, (H + 15) // 16
            empty = (R.bin_sections(st)["tile_count"].reshape(n_views, gy, gx) == 0)
            empty_px = empty.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W]
            frac_empty = float(empty.float().mean())
            if poison:
                n = n_views * H * W * 4
                off = (n + 255) & ~255
                st.img[:n].view(torch.float32).reshape(n_views, H, W)[empty_px] = float("nan")
                st.img[off:off + n].view(torch.int32).reshape(n_views, H, W)[empty_px] = 0x7fffffff
            count = torch.zeros(1, dtype=torch.int64, device=dev)
            _lib.check(_lib.lib().ggs_count_blends(C.byref(st.prm), st.geom.data_ptr(), st.bin.data_ptr(), st.cap, st.img.data_ptr(),
                                                   count.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "ggs_count_blends")
            g = R.backward_views(st, dL, want_means2D=True)
        return {k: t.clone() for k, t in g.items() if torch.is_tensor(t)}, int(count.item()), frac_empty

    clean, n_clean, frac_empty = run(False)
    dirty, n_dirty, _ = run(True)
    assert frac_empty > 0.3 and n_clean > 0 and n_dirty == n_clean
    for k in clean:
        assert torch.isfinite(dirty[k]).all(), k
        assert rel_l1(dirty[k], clean[k]) <= 1e-6, k


def test_host_resident_camera_tensors_are_moved_not_dereferenced():
    """Camera matrices left on the host (synthetic.rig_cameras() builds them there) used to reach the kernels as host pointers:
    a GPU memory fault.  They are moved to the device like the background colour always was; same image either way."""
    from types import SimpleNamespace
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.render import render
    v, f = S.skirt_mesh(20, 30)
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], 0), 0, device="cuda")
    cam = S.rig_cameras(n_rings=1, n_az=4, width=160, height=120, f=130.0)[2]
    assert cam.world_view_transform.device.type == "cpu"
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    with torch.no_grad():
        m.update_face_coor()
        on_host = render(cam, m, pipe, torch.zeros(3))["render"].clone()             # background on the host too
        for n in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(cam, n, getattr(cam, n).cuda())
        on_dev = render(cam, m, pipe, torch.zeros(3, device="cuda"))["render"]
    assert float(on_host.abs().sum()) > 0 and torch.equal(on_host, on_dev)


def test_backward_twice_on_one_forward_gives_the_same_gradients():
    """A second backward over the same forward state (retain_graph=True) reproduces the first up to the order of the float
    atomics: the per-(view, Gaussian) accumulators are cleared by every ggs_backward call, nothing of the first call leaks into
    the second.  Single-view and batched calls, with and without depth / alpha gradients."""
    from ggsplat import rasterizer as R
    dev = "cuda"
    sc, _ = small_scene(P=900, sh_degree=1, seed=11)
    cams = S.orbit_cameras(3, width=112, img_height=80, fx=90.0, fy=90.0, cx=55.0, cy=41.0)
    ck = S.stack_cameras(cams, device=dev)
    inp = [sc[k].to(dev) for k in ("means3D", "opacities", "shs")] + [None] + [sc[k].to(dev) for k in ("scales", "rotations")] + [None]
    g = torch.Generator().manual_seed(4)
    for V in (1, 3):
        color, radii, depth, alpha, st = R.forward_views(
            *inp, view=ck["view"][:V], proj=ck["proj"][:V], campos=ck["campos"][:V], tanfov=ck["tanfov"][:V],
            bg=torch.tensor([0.2, 0.1, 0.4], device=dev), W=112, H=80, sh_degree=1)
        dL = torch.randn(V, 3, 80, 112, generator=g).to(dev)
        dD, dA = torch.randn(V, 80, 112, generator=g).to(dev), torch.randn(V, 80, 112, generator=g).to(dev)
        for extra in ((None, None), (dD, dA)):
            first = {k: v.clone() for k, v in R.backward_views(st, dL, extra[0], extra[1], want_means2D=True).items()}
            second = R.backward_views(st, dL, extra[0], extra[1], want_means2D=True)
            assert float(first["means3D"].abs().sum()) > 0
            for k in first:
                assert rel_l1(second[k], first[k]) <= 1e-6, (V, k)


def test_forward_stages_queued_one_by_one_equal_the_one_call_forward():
    """ggs_forward_stages (round 5): COUNT, BIN and COMPOSITE queued as three calls (rasterizer.StagedForward) give the outputs,
    the per-pixel workspace and the sorted lists of the one-call forward bit for bit, a staged forward feeds the backward like any
    other, and a stage mask that is not a combination of the three stages is an argument error, not a launch."""
    import ctypes as C
    from ggsplat import _lib, rasterizer as R
    sc, cam = small_scene(P=900, W=112, H=80, sh_degree=1, seed=21)
    dev = "cuda"
    t = {k: sc[k].to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    cams = S.stack_cameras([cam, cam], device=dev)
    kw = dict(view=cams["view"], proj=cams["proj"], campos=cams["campos"], tanfov=cams["tanfov"],
              bg=torch.tensor([0.1, 0.2, 0.3], device=dev), W=112, H=80, sh_degree=1)
    color, radii, depth, alpha, st = R.forward_views(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None, **kw)
    fwd = R.StagedForward(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None, **kw)
    for stage in (fwd.COUNT, fwd.BIN, fwd.COMPOSITE):
        fwd.run(stage)
    torch.cuda.synchronize()
    c2, r2, d2, a2 = fwd.outputs
    assert torch.equal(c2, color) and torch.equal(r2, radii) and torch.equal(d2, depth) and torch.equal(a2, alpha)
    assert int(fwd.header[1]) == 0 and int(fwd.header[0]) == st.num_rendered
    s1, s2 = R.img_sections(st), R.img_sections(fwd.state)
    assert torch.equal(s1["final_T"], s2["final_T"]) and torch.equal(s1["n_contrib"], s2["n_contrib"])
    n = st.num_rendered
    assert torch.equal(R.bin_sections(st)["ids"][:n], R.bin_sections(fwd.state)["ids"][:n])
    w = torch.randn(2, 3, 80, 112, generator=torch.Generator().manual_seed(4)).to(dev)
    g1, g2 = R.backward_views(st, w), R.backward_views(fwd.state, w)
    for k in g1:
        assert rel_l1(g2[k], g1[k]) <= 2e-6, k
    L = _lib.lib()
    for bad in (0, 8, -1, 16 | 1, fwd.COUNT | fwd.COMPOSITE):      # the last: not a contiguous run
        assert L.ggs_forward_stages(bad, *fwd._args, _lib.stream_ptr(torch.device(dev))) != 0
        assert b"ggs_forward_stages" in L.ggs_last_error()
    # a shape that was never run through forward_views has no learnt binning capacity: refused, not guessed
    R._cap_hint.pop((torch.device(dev).index or 0, 900, 112, 80, 1), None)
    with pytest.raises(_lib.GgsError, match="forward_views once"):
        R.StagedForward(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None,
                        **{**kw, **{k: v[:1] for k, v in cams.items()}})


def test_loss_callback_sees_the_tables_of_its_own_set_in_every_pipeline_mode():
    """rasterizer.last_tile_count() / last_header() inside dL_dcolor_fn describe the forward whose images the callback was
    handed -- serial, pipeline=1 and the staged form (pipeline=2), where the NEXT set's count + bin stages are already queued
    when the callback runs (the region-of-interest loss would zero gradients in the wrong tiles otherwise)."""
    from ggsplat import batch, rasterizer as R
    sc, _ = small_scene(P=700, W=96, H=64, sh_degree=0, seed=5)
    cams_l = S.orbit_cameras(6, width=96, img_height=64, fx=105.0, fy=105.0, cx=47.0, cy=33.0)
    dev = "cuda"
    t = {k: sc[k].to(dev) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    cams = S.stack_cameras(cams_l, device=dev)
    bg = torch.zeros(3, device=dev)
    w = torch.randn(6, 3, 64, 96, generator=torch.Generator().manual_seed(9)).to(dev)
    want = {}
    for v0 in range(0, 6, 2):       # per launch set of two views: the list lengths of ITS forward
        *_, st = R.forward_views(t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None,
                                 view=cams["view"][v0:v0 + 2], proj=cams["proj"][v0:v0 + 2], campos=cams["campos"][v0:v0 + 2],
                                 tanfov=cams["tanfov"][v0:v0 + 2], bg=bg, W=96, H=64, sh_degree=0)
        want[v0] = (R.bin_sections(st)["tile_count"].clone(), st.num_rendered)
    assert not torch.equal(want[0][0], want[2][0])
    for pipeline in (0, 1, 2):
        seen = {}

        def fn(v0, v1, color):
            seen[v0] = (R.last_tile_count().clone(), R.last_header().clone())
            return w[v0:v1]
        batch.fwd_bwd_views(t, cams, bg=bg, W=96, H=64, sh_degree=0, dL_dcolor_fn=fn, chunk=2, pipeline=pipeline)
        torch.cuda.synchronize()
        assert sorted(seen) == [0, 2, 4], pipeline
        for v0, (tc, hdr) in seen.items():
            assert torch.equal(tc, want[v0][0]), (pipeline, v0)
            assert int(hdr[0]) == want[v0][1] and int(hdr[1]) == 0, (pipeline, v0)
