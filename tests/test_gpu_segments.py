"""Segment-parallel compositing (csrc/ggs_render_seg.hip) -- the latency mapping of small launches, where a tile list is
walked by several waves (s2_registration.py:241-251 renders ONE view per optimisation step).

  * against the unsegmented kernels on the same inputs: images, depth, alpha, final_T to fp32 rounding (the products /
    sums are re-associated across segment boundaries), n_contrib and radii exactly, every gradient <= 1e-5;
  * against the C oracle with short segments (64) on a scene with long lists, early termination and ragged image size;
  * at full size (100k Gaussians, 1080p, one view) with the default segment length: against the C oracle.
"""
import math

import numpy as np
import pytest
import torch

from helpers import REL_L1_TOL, cam_kwargs, rel_l1, seeded_image_weights, small_scene
from ggsplat import synthetic as S
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_segment_length():
    from ggsplat import rasterizer as R
    yield
    R.set_segment_length(-1)


def _run(sc, cam, bg, seg_len, weights, da=True):
    from ggsplat import rasterizer as R
    from ggsplat.synthetic import stack_cameras
    R.set_segment_length(seg_len)
    dev = "cuda"
    ck = stack_cameras([cam], device=dev)
    W, H = cam.image_width, cam.image_height
    color, radii, depth, alpha, st = R.forward_views(
        sc["means3D"].to(dev), sc["opacities"].to(dev), sc["shs"].to(dev), None, sc["scales"].to(dev),
        sc["rotations"].to(dev), None, view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"],
        bg=torch.tensor(bg, device=dev), W=W, H=H, sh_degree=sc["sh_degree"])
    sec = R.img_sections(st)
    wc, wd, wa = (w.to(dev) for w in weights)
    g = R.backward_views(st, wc[None], wd if da else None, wa if da else None, want_means2D=True)
    out = dict(color=color[0], depth=depth[0], alpha=alpha[0], radii=radii[0], final_T=sec["final_T"][0].clone(),
               n_contrib=sec["n_contrib"][0].clone(), tile_max=int(R.bin_sections(st)["tile_count"].max()))
    out.update({"d" + k: v.clone() for k, v in g.items()})
    return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}


@pytest.mark.parametrize("seg_len,da", [(64, True), (128, False), (192, True)])
def test_segmented_equals_unsegmented(seg_len, da):
    sc, cam = small_scene(P=6000, W=150, H=101, sh_degree=1, seed=41, scale_mul=9.0, opacity_boost=1.0, cam_index=2)
    weights = seeded_image_weights(cam.image_width, cam.image_height)
    bg = (0.3, 0.1, 0.7)
    ref = _run(sc, cam, bg, 0, weights, da)
    seg = _run(sc, cam, bg, seg_len, weights, da)
    assert ref["tile_max"] > 3 * seg_len                                   # lists really are cut into several segments
    assert float(ref["final_T"].min()) < 2e-4                              # ... and pixels terminate inside them
    assert torch.equal(seg["radii"], ref["radii"]) and torch.equal(seg["n_contrib"], ref["n_contrib"])
    for k in ("color", "depth", "alpha", "final_T"):
        assert rel_l1(seg[k], ref[k]) <= 2e-6, k
    for k in ref:
        if k.startswith("d"):
            assert rel_l1(seg[k], ref[k]) <= 1e-5, k


def test_segmented_against_the_c_oracle():
    sc, cam = small_scene(P=5000, W=131, H=77, sh_degree=2, seed=43, scale_mul=8.0, opacity_boost=2.0, cam_index=1)
    weights = seeded_image_weights(cam.image_width, cam.image_height)
    bg = (0.9, 0.2, 0.4)
    out = _run(sc, cam, bg, 64, weights)
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=2, **cam_kwargs(cam, bg))
    og = co.backward(*weights)
    assert np.array_equal(out["radii"].numpy(), co.radii)
    assert rel_l1(out["color"], co.color) <= REL_L1_TOL and rel_l1(out["depth"], co.depth.reshape(out["depth"].shape)) <= REL_L1_TOL
    assert rel_l1(out["alpha"], co.alpha.reshape(out["alpha"].shape)) <= REL_L1_TOL
    assert rel_l1(out["final_T"], co.internals()["final_T"]) <= REL_L1_TOL
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_l1(out["d" + k].reshape(og[k].shape), og[k]) <= REL_L1_TOL, k
    assert rel_l1(out["dmeans2D"][0].reshape(og["means2D"].shape), og["means2D"]) <= REL_L1_TOL


def test_segmented_full_size_view_against_the_c_oracle():
    """BASELINE size, one view (the shape of the reference's per-step render), 128 entries per segment."""
    from ggsplat import rasterizer as R
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    W, H = 1920, 1080
    v, f = S.skirt_mesh()
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], sh_degree=0), sh_degree=0, device="cuda")
    m.update_face_coor()
    with torch.no_grad():
        sc = dict(means3D=m.get_xyz.cpu(), scales=m.get_scaling.cpu(), rotations=m.get_rotation.cpu(),
                  opacities=m.get_opacity.cpu(), shs=m.get_features.cpu(), sh_degree=0)
    cam = S.rig_cameras()[101]
    g = torch.Generator().manual_seed(7)
    weights = (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g) * 0.3, torch.randn(1, H, W, generator=g))
    out = _run(sc, cam, (0.0, 0.0, 0.0), 128, weights)
    assert out["tile_max"] > 256                                             # several 128-entry segments per heavy tile
    co = COracle(means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"],
                 rotations=sc["rotations"], sh_degree=0, **cam_kwargs(cam, (0.0, 0.0, 0.0)))
    og = co.backward(*weights)
    assert np.array_equal(out["radii"].numpy(), co.radii)
    assert rel_l1(out["color"], co.color) <= REL_L1_TOL and rel_l1(out["alpha"], co.alpha.reshape(H, W)) <= REL_L1_TOL
    assert rel_l1(out["depth"], co.depth.reshape(H, W)) <= REL_L1_TOL
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_l1(out["d" + k].reshape(og[k].shape), og[k]) <= REL_L1_TOL, k
