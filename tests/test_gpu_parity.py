"""GPU parity: HIP path (through the C ABI, via the drop-in module) vs the C oracle.

Bars (north_star): bit-exact for integer / index outputs -- radii; the per-tile sorted id lists (exact subsequences of
the oracle's lists, every dropped pair proven unblendable); the Gaussian id of every pixel's last contributor (the
per-pixel counts themselves are positions in differently culled lists and are compared through that id) -- and
<= 1e-4 relative L1 for fp32 images and gradients (tolerance REL_L1_TOL in helpers.py).
"""
import math

import numpy as np
import pytest
import torch

from helpers import REL_L1_TOL, cam_kwargs, rel_l1, seeded_image_weights, small_scene
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


def _settings(cam, bg, sh_degree, dev, scale_modifier=1.0, debug=False):
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.as_tensor(bg, dtype=torch.float32, device=dev),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=sh_degree, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=debug)


def _run_hip(sc, cam, bg, mode, weights, dev="cuda", scale_modifier=1.0):
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizer
    from oracle import torch_oracle as TO
    leaf = {}

    def L(name, t):
        leaf[name] = t.clone().to(dev).requires_grad_(True)
        return leaf[name]

    P = sc["means3D"].shape[0]
    kw = dict(means3D=L("means3D", sc["means3D"]), means2D=L("means2D", torch.zeros(P, 3)),
              opacities=L("opacities", sc["opacities"]))
    if mode["sh"]:
        kw["shs"] = L("shs", sc["shs"])
    else:
        kw["colors_precomp"] = L("colors_precomp", sc["colors"])
    if mode["cov"]:
        kw["cov3D_precomp"] = L("cov3D_precomp", sc["cov"])
    else:
        kw["scales"] = L("scales", sc["scales"])
        kw["rotations"] = L("rotations", sc["rotations"])
    rast = GaussianRasterizer(_settings(cam, bg, sc["sh_degree"], dev, scale_modifier))
    color, radii, depth, alpha = rast(**kw)
    wc, wd, wa = (w.to(dev) for w in weights)
    loss = (color * wc).sum()
    if mode.get("da", True):
        loss = loss + (depth * wd).sum() + (alpha * wa).sum()
    loss.backward()
    grads = {k: v.grad.detach().cpu() for k, v in leaf.items() if v.grad is not None}
    return color.detach().cpu(), radii.cpu(), depth.detach().cpu(), alpha.detach().cpu(), grads


def _run_oracle(sc, cam, bg, mode, weights, scale_modifier=1.0):
    kw = cam_kwargs(cam, bg)
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"] if mode["sh"] else None,
                 colors_precomp=None if mode["sh"] else sc["colors"],
                 scales=None if mode["cov"] else sc["scales"], rotations=None if mode["cov"] else sc["rotations"],
                 cov3D_precomp=sc["cov"] if mode["cov"] else None, sh_degree=sc["sh_degree"],
                 scale_modifier=scale_modifier, **kw)
    wc, wd, wa = weights
    g = co.backward(wc, wd if mode.get("da", True) else None, wa if mode.get("da", True) else None)
    return co, g


def _add_precomp(sc):
    from oracle import torch_oracle as TO
    g = torch.Generator().manual_seed(99)
    sc["cov"] = TO.cov3d_from_scale_rot(sc["scales"], 1.0, sc["rotations"]).contiguous()
    sc["colors"] = torch.rand(sc["means3D"].shape[0], 3, generator=g)
    return sc


def _compare(sc, cam, bg, mode, scale_modifier=1.0):
    weights = seeded_image_weights(cam.image_width, cam.image_height)
    color, radii, depth, alpha, grads = _run_hip(sc, cam, bg, mode, weights, scale_modifier=scale_modifier)
    co, og = _run_oracle(sc, cam, bg, mode, weights, scale_modifier=scale_modifier)
    assert np.array_equal(radii.numpy(), co.radii), "radii must be bit-exact"
    assert rel_l1(color, co.color) <= REL_L1_TOL
    assert rel_l1(depth, co.depth) <= REL_L1_TOL
    assert rel_l1(alpha, co.alpha) <= REL_L1_TOL
    names = {"means3D": "means3D", "means2D": "means2D", "opacities": "opacities", "shs": "shs",
             "colors_precomp": "colors", "cov3D_precomp": "cov3D", "scales": "scales", "rotations": "rotations"}
    for k, t in grads.items():
        ref = og[names[k]]
        assert rel_l1(t.reshape(ref.shape), ref) <= REL_L1_TOL, f"grad {k}"
    return co


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_scale_rot_path(deg):
    sc, cam = small_scene(P=700, W=96, H=80, sh_degree=deg, seed=10 + deg)
    _compare(sc, cam, (0.2, 0.5, 0.7), dict(sh=True, cov=False))


def test_precomp_paths_and_clamp_termination():
    sc, cam = small_scene(P=500, W=70, H=50, sh_degree=1, seed=5, scale_mul=25.0, opacity_boost=3.0, cam_index=2)
    sc = _add_precomp(sc)
    co = _compare(sc, cam, (0.9, 0.1, 0.4), dict(sh=False, cov=True))
    it = co.internals()
    assert it["final_T"].min() < 2e-4          # early termination actually exercised


def test_no_depth_alpha_grads_kernel():
    sc, cam = small_scene(P=800, W=64, H=64, sh_degree=0, seed=21, scale_mul=6.0)
    _compare(sc, cam, (0.0, 0.0, 0.0), dict(sh=True, cov=False, da=False))


def _segments_scheduled(sc, cam, dev="cuda"):
    """How the segmented backward cuts this view's lists: (lists of >= 64 entries, longest list, n_extra) -- ggs_seg_item of
    csrc/ggs_common.h replayed on the forward's bin header."""
    from ggsplat import rasterizer as R
    from ggsplat.synthetic import stack_cameras
    cams = stack_cameras([cam], device=dev)
    *_, st = R.forward_views(sc["means3D"].to(dev), sc["opacities"].to(dev), None, sc["colors"].to(dev), sc["scales"].to(dev),
                             sc["rotations"].to(dev), None, view=cams["view"], proj=cams["proj"], campos=cams["campos"],
                             tanfov=cams["tanfov"], bg=torch.zeros(3, device=dev), W=cam.image_width, H=cam.image_height, sh_degree=0)
    counts = R.bin_sections(st)["tile_count"].cpu().numpy().reshape(-1)
    bucket = st.bin[64:128].view(torch.int32).cpu().numpy().astype(np.int64)
    T = counts.size
    n_ne = int((counts > 0).sum())
    assert int(bucket[15]) == T - n_ne
    lenlo = [3072, 2048, 1536, 1024, 768, 512, 384, 256, 192, 128, 96, 64]
    cum = np.cumsum(bucket[:12])
    n_extra, offset, c = 0, 0, 11
    for k in range(1, 24):
        while c > 0 and lenlo[c - 1] <= k * 64:
            c -= 1
        if cum[c] == 0 or offset + cum[c] > T - n_ne:
            break
        n_extra, offset = k, offset + int(cum[c])
    return int(cum[11]), int(counts.max()), n_extra


@pytest.mark.parametrize("P,spread,W,H,regime", [(1000, 0.12, 320, 256, "all"), (2000, 0.30, 128, 96, "truncated"), (2600, 1.0, 64, 48, "none")])
def test_segmented_backward_in_all_scheduling_regimes(P, spread, W, H, regime):
    """The latency-mapped backward without depth / alpha gradients walks SEGMENTS of 64 list positions from the forward's
    checkpoints (csrc/ggs_common.h GGS_SEG); the later segments ride on the blocks of empty tiles.  Three regimes against the
    C oracle, every gradient <= 1e-4: a cluster in a mostly empty image (every segment of every list has its own wave), fewer
    spare blocks than segments (the last scheduled segment is open-ended), and an image without empty tiles (one wave per list)."""
    sc, cam = small_scene(P=P, W=W, H=H, sh_degree=0, seed=17, scale_mul=2.0)
    sc["means3D"] = sc["means3D"] * spread                       # a cluster around the look-at point: long lists on few tiles
    sc = _add_precomp(sc)
    n_long, longest, n_extra = _segments_scheduled(sc, cam)
    needed = (longest - 1) // 64
    assert longest > 192, longest
    if regime == "all":
        assert n_extra >= needed >= 3, (n_extra, needed)
    elif regime == "truncated":
        assert 1 <= n_extra < needed, (n_extra, needed)
    else:
        assert n_extra == 0 and n_long > 0
    for bg in ((0.0, 0.0, 0.0), (0.3, 0.6, 0.1)):
        _compare(sc, cam, bg, dict(sh=False, cov=False, da=False))


def test_scale_modifier():
    sc, cam = small_scene(P=400, W=64, H=48, sh_degree=0, seed=23, scale_mul=5.0)
    _compare(sc, cam, (1.0, 1.0, 1.0), dict(sh=True, cov=False), scale_modifier=0.7)


def test_internals_bit_exact():
    """Per-Gaussian records bit-exact; per-tile sorted lists are exact subsequences of the oracle lists."""
    from ggsplat import rasterizer as R
    from ggsplat.synthetic import stack_cameras
    sc, cam = small_scene(P=3000, W=160, H=120, sh_degree=0, seed=31, scale_mul=5.0)
    dev = "cuda"
    cams = stack_cameras([cam], device=dev)
    color, radii, depth, alpha, st = R.forward_views(
        sc["means3D"].to(dev), sc["opacities"].to(dev), sc["shs"].to(dev), None, sc["scales"].to(dev),
        sc["rotations"].to(dev), None, view=cams["view"], proj=cams["proj"], campos=cams["campos"],
        tanfov=cams["tanfov"], bg=torch.zeros(3, device=dev), W=160, H=120, sh_degree=0)
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=0, **cam_kwargs(cam, (0, 0, 0)))
    it = co.internals()
    final_T = R.img_sections(st)["final_T"][0].cpu()       # (empty tiles: 1 by definition, the kernels do not store them)
    assert rel_l1(final_T, it["final_T"]) <= REL_L1_TOL
    assert rel_l1(color[0], co.color) <= REL_L1_TOL and rel_l1(alpha, co.alpha) <= REL_L1_TOL
    # Binning: the HIP lists may DROP (splat, tile) pairs whose alpha can never reach 1/255 inside the tile
    # (conservative AABB culling, output-invariant), never add or reorder: each tile's sorted id list must be
    # a subsequence of the oracle's list, and every dropped pair must be one no pixel of the tile blends.
    from ggsplat.rasterizer import bin_sections
    sec = bin_sections(st)
    counts = sec["tile_count"].cpu().numpy().astype(np.int64).reshape(-1)
    offs = sec["tile_offset"].cpu().numpy().astype(np.int64).reshape(-1)
    ids = sec["ids"].cpu().numpy().astype(np.uint32) & np.uint32(0x0fffffff)   # bits 28..31: quadrant mask
    assert st.num_rendered == counts.sum() <= co.num_rendered
    assert st.num_rendered > 0.5 * co.num_rendered
    xy, con = it["xy"], it["conic_opacity"]
    gx = 10
    for t in range(len(counts)):
        mine = ids[offs[t]:offs[t] + counts[t]]
        ref = it["list"][it["tile_start"][t]:it["tile_start"][t + 1]]
        pos = {int(g): i for i, g in enumerate(ref)}
        idx = [pos[int(g)] for g in mine]                    # KeyError = an id the oracle does not have
        assert idx == sorted(idx), f"tile {t}: order differs"
        dropped = sorted(set(int(g) for g in ref) - set(int(g) for g in mine))
        if dropped:
            ys, xs = np.meshgrid(np.arange((t // gx) * 16, (t // gx) * 16 + 16),
                                 np.arange((t % gx) * 16, (t % gx) * 16 + 16), indexing="ij")
            for g in dropped:
                dx, dy = xy[g, 0] - xs, xy[g, 1] - ys
                power = -0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) - con[g, 1] * dx * dy
                a_ = np.where(power > 0, 0.0, np.minimum(0.99, con[g, 3] * np.exp(np.minimum(power, 0))))
                assert a_.max() < 1.0 / 255.0, f"tile {t}: dropped splat {g} would have been blended"
    # Index output of the compositing: the LAST CONTRIBUTOR of every pixel.  n_contrib is a 1-based position in the
    # tile's list -- the culled list here, the full list in the oracle -- so the comparable quantity is the Gaussian id
    # found at that position: it must be the same Gaussian, pixel for pixel (and 0 contributors in the same pixels).
    n_hip = R.img_sections(st)["n_contrib"][0].cpu().numpy().astype(np.int64)
    n_ref = it["n_contrib"].astype(np.int64)
    ty, tx = np.meshgrid(np.arange(120) // 16, np.arange(160) // 16, indexing="ij")
    tile = ty * gx + tx
    assert np.array_equal(n_hip > 0, n_ref > 0)
    has = n_ref > 0
    last_hip = ids[(offs[tile] + n_hip - 1)[has]].astype(np.int64)
    last_ref = it["list"][(it["tile_start"][tile] + n_ref - 1)[has]].astype(np.int64)
    assert np.array_equal(last_hip, last_ref), f"{int((last_hip != last_ref).sum())} of {int(has.sum())} pixels end on another Gaussian"
    assert has.sum() > 0.3 * 160 * 120
    # per-Gaussian records: bit-exact geometry (floats 0..9 of the 48-byte record)
    rec = st.geom.cpu()[:3000 * 48].view(torch.float32).reshape(3000, 12).numpy()
    vis = co.radii > 0
    assert np.array_equal(rec[vis, 0:2], it["xy"][vis])
    assert np.array_equal(rec[vis, 9], it["depth"][vis])
    # the record stores the conic pre-scaled for the compositing loops: (-log2e/2, -log2e, -log2e/2) * conic
    k = np.array([-0.72134752044448170, -1.44269504088896340, -0.72134752044448170], dtype=np.float32)
    assert np.array_equal(rec[vis][:, [2, 3, 4]], (it["conic_opacity"][vis][:, :3] * k[None, :]).astype(np.float32))
    assert np.array_equal(rec[vis][:, 5], it["conic_opacity"][vis][:, 3])
    assert np.array_equal(rec[vis][:, [6, 7, 8]], it["rgb"][vis])


def test_empty_and_offscreen():
    """P = 0 and all-culled inputs: background image, zero alpha, no crash."""
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizer
    sc, cam = small_scene(P=50, W=48, H=32, sh_degree=0, seed=2)
    dev = "cuda"
    bg = (0.3, 0.6, 0.9)
    rast = GaussianRasterizer(_settings(cam, bg, 0, dev))
    behind = sc["means3D"].to(dev) * 0 + cam.camera_center.to(dev) - 5.0 * torch.tensor(cam.R[:, 2], dtype=torch.float32, device=dev)
    color, radii, depth, alpha = rast(means3D=behind, means2D=torch.zeros(50, 3, device=dev), opacities=sc["opacities"].to(dev),
                                      shs=sc["shs"].to(dev), scales=sc["scales"].to(dev), rotations=sc["rotations"].to(dev))
    assert int((radii > 0).sum()) == 0
    assert torch.allclose(color, torch.tensor(bg, device=dev)[:, None, None].expand_as(color))
    assert float(alpha.abs().max()) == 0.0
    e = torch.zeros(0, 3, device=dev)
    color, radii, depth, alpha = rast(means3D=e, means2D=e, opacities=torch.zeros(0, 1, device=dev),
                                      shs=torch.zeros(0, 1, 3, device=dev), scales=e, rotations=torch.zeros(0, 4, device=dev))
    assert radii.numel() == 0 and torch.allclose(color, torch.tensor(bg, device=dev)[:, None, None].expand_as(color))


def test_argument_errors():
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizer
    sc, cam = small_scene(P=10, W=32, H=32, sh_degree=0)
    rast = GaussianRasterizer(_settings(cam, (0, 0, 0), 0, "cuda"))
    d = {k: v.cuda() for k, v in sc.items() if torch.is_tensor(v)}
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=d["means3D"], means2D=torch.zeros(10, 3).cuda(), opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=d["means3D"], means2D=torch.zeros(10, 3).cuda(), opacities=d["opacities"], shs=d["shs"])


def test_multi_view_batch_matches_single_views():
    """V views in one launch == V single-view calls; summed gradients == sum of per-view gradients."""
    from ggsplat import rasterizer as R
    from ggsplat import synthetic as S
    dev = "cuda"
    sc = S.random_gaussians(1500, sh_degree=2, seed=8)
    sc["scales"] *= 5
    cams = S.orbit_cameras(4, width=96, img_height=64, fx=100., fy=100., cx=47., cy=33.)
    ck = S.stack_cameras(cams, device=dev)
    args = [sc["means3D"].to(dev), sc["opacities"].to(dev), sc["shs"].to(dev), None, sc["scales"].to(dev), sc["rotations"].to(dev), None]
    common = dict(bg=torch.tensor([0.1, 0.2, 0.3], device=dev), W=96, H=64, sh_degree=2)
    color, radii, depth, alpha, st = R.forward_views(*args, view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], **common)
    g = torch.Generator().manual_seed(4)
    dc = torch.randn(4, 3, 64, 96, generator=g).to(dev)
    gb = R.backward_views(st, dc)
    acc = None
    for v in range(4):
        c1, r1, d1, a1, s1 = R.forward_views(*args, view=ck["view"][v:v + 1], proj=ck["proj"][v:v + 1], campos=ck["campos"][v:v + 1],
                                             tanfov=ck["tanfov"][v:v + 1], **common)
        assert torch.equal(c1[0], color[v]) and torch.equal(r1[0], radii[v]) and torch.equal(d1[0], depth[v])
        g1 = R.backward_views(s1, dc[v:v + 1])
        assert rel_l1(g1["means2D"][0], gb["means2D"][v]) <= 1e-6
        if acc is None:
            acc = {k: t.clone() for k, t in g1.items() if k != "means2D"}
        else:
            for k in acc:
                acc[k] += g1[k]
    for k in acc:
        assert rel_l1(gb[k], acc[k]) <= 1e-5, k


# (P, W, H, sh_degree, seed, scale_mul, opacity_boost, cam, bg): ragged image sizes (not multiples of the 16-px tile
# or the 8-px quadrant), one / a handful of splats, splats spanning hundreds of tiles (rect > 64 tiles: the
# reachable-tile bitmask is not used), near-opaque splats (0.99 clamp, early termination), non-zero backgrounds
SWEEP = [
    (1, 33, 17, 0, 5, 30.0, 4.0, 0, (0.2, 0.5, 0.9)),
    (7, 77, 53, 1, 6, 12.0, 2.0, 1, (0.0, 0.0, 0.0)),
    (300, 130, 95, 2, 7, 3.0, 0.0, 2, (1.0, 1.0, 1.0)),
    (900, 208, 120, 3, 8, 6.0, 3.0, 3, (0.3, 0.1, 0.7)),
    (2500, 161, 161, 0, 9, 1.5, 1.0, 1, (0.0, 0.0, 0.0)),
    (64, 320, 200, 1, 10, 40.0, -1.0, 2, (0.5, 0.5, 0.5)),
]


@pytest.mark.parametrize("P,W,H,deg,seed,scale_mul,boost,cam_index,bg", SWEEP)
def test_random_sweep(P, W, H, deg, seed, scale_mul, boost, cam_index, bg):
    sc, cam = small_scene(P=P, W=W, H=H, sh_degree=deg, seed=seed, scale_mul=scale_mul, cam_index=cam_index,
                          opacity_boost=boost)
    _compare(_add_precomp(sc), cam, list(bg), dict(sh=True, cov=False))
    if seed % 2 == 0:
        _compare(sc, cam, list(bg), dict(sh=False, cov=True, da=False))


@pytest.mark.parametrize("cam_index,bg", [(0, (0.0, 0.0, 0.0)), (1, (0.7, 0.2, 0.4)), (2, (0.0, 0.0, 0.0)), (3, (0.1, 0.9, 0.5))])
def test_config1_plumbing_case(cam_index, bg):
    """BASELINE config 1 as specified: 10k random Gaussians, SH degree 3, 4 cameras on a radius-4 circle,
    512x512, fx = fy = 600, principal point (250, 262), black and non-black backgrounds."""
    sc = S_random(10_000)
    cam = S_orbit()[cam_index]
    _compare(_add_precomp(sc), cam, list(bg), dict(sh=True, cov=False))


def S_random(P):
    from ggsplat import synthetic as S
    return S.random_gaussians(P, sh_degree=3, seed=0)


def S_orbit():
    from ggsplat import synthetic as S
    return S.orbit_cameras(4)


@pytest.mark.parametrize("tag,mod", [("1", 1.0), ("1p3", 1.3)])
def test_cov3d_stage_against_the_reference_fixture(tag, mod):
    """The scale/rotation -> cov3D stage of the HIP preprocess, pinned to tests/golden/cov3d.npz (generated from the
    reference's build_scaling_rotation / strip_symmetric, utils/general_utils.py:91-120): rendering from (scales,
    rotations, scale_modifier) must equal rendering from cov3D_precomp = the reference's own Sigma -- same radii, same
    image, same depth / alpha."""
    import os
    from ggsplat import rasterizer as R
    from ggsplat import synthetic as S
    from ggsplat.synthetic import stack_cameras
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cov3d.npz"))
    dev = "cuda"
    s = torch.tensor(d["scales"]).to(dev)
    qn = torch.nn.functional.normalize(torch.tensor(d["rots"])).to(dev)
    cov = torch.tensor(d[f"cov6_{tag}"]).to(dev)
    P = s.shape[0]
    g = torch.Generator().manual_seed(1)
    means = (torch.randn(P, 3, generator=g) * 0.3).to(dev)
    cols = torch.rand(P, 3, generator=g).to(dev)
    op = torch.full((P, 1), 0.6, device=dev)
    cam = S.orbit_cameras(4, width=96, img_height=64, fx=110.0, fy=110.0, cx=47.0, cy=33.0)[0]
    ck = stack_cameras([cam], device=dev)
    kw = dict(view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev),
              W=96, H=64, sh_degree=0, keep_state=False)
    a = R.forward_views(means, op, None, cols, s, qn, None, scale_modifier=mod, **kw)
    b = R.forward_views(means, op, None, cols, None, None, cov, **kw)
    assert int((a[1] > 0).sum()) > P // 2 and torch.equal(a[1], b[1])
    for x, y in zip((a[0], a[2], a[3]), (b[0], b[2], b[3])):
        assert rel_l1(x, y) <= 1e-5


@pytest.mark.parametrize("views", [1, 8])
def test_count_blends_equals_the_oracles_blended_pairs(views):
    """ggs_count_blends (the basis of bench.py's compute-side roofline) against the C oracle's own count of the
    (Gaussian, pixel) pairs its compositing loop blended: one view through the per-quadrant kernels, eight through the
    per-tile kernels.  Exact up to pairs whose alpha sits within rounding of 1/255 (exp2 on the GPU, expf in the oracle)."""
    import ctypes as C
    from ggsplat import _lib, rasterizer as R
    from ggsplat.synthetic import orbit_cameras, random_gaussians, stack_cameras
    sc = random_gaussians(4000, sh_degree=0, seed=77)
    sc["scales"] = sc["scales"] * 4
    cams = orbit_cameras(views, width=1920 if views > 1 else 320, img_height=1080 if views > 1 else 200,
                         fx=900.0 if views > 1 else 220.0, fy=900.0 if views > 1 else 220.0,
                         cx=960.0 if views > 1 else 158.0, cy=540.0 if views > 1 else 101.0)
    W, H = (1920, 1080) if views > 1 else (320, 200)
    dev = "cuda"
    ck = stack_cameras(cams, device=dev)
    color, radii, depth, alpha, st = R.forward_views(
        sc["means3D"].to(dev), sc["opacities"].to(dev), sc["shs"].to(dev), None, sc["scales"].to(dev),
        sc["rotations"].to(dev), None, view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"],
        bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().ggs_count_blends(C.byref(st.prm), st.geom.data_ptr(), st.bin.data_ptr(), st.cap, st.img.data_ptr(),
                                           cnt.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ggs_count_blends")
    ref = 0
    for cam in cams:
        co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                     rotations=sc["rotations"], sh_degree=0, **cam_kwargs(cam, (0, 0, 0)))
        ref += co.num_blended
        co.close()
    got = int(cnt.item())
    assert ref > 10000 and abs(got - ref) <= max(2, int(2e-5 * ref)), (got, ref)
