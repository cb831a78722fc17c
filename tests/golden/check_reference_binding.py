#!/usr/bin/env python
"""The reference's OWN gaussian_renderer/__init__.py on top of the REAL drop-in module.

make_golden.py records what the reference's render() / doll_render() hand a STAND-IN rasterizer module
(render_args.npz), and tests/test_golden.py checks this repo's mirror (ggsplat/render.py) against that
recording.  Neither puts the reference's file on top of gaussian-garments_amd/diff_gaussian_rasterization_depth_alpha
itself.  This script does: /root/reference/gaussian_renderer/__init__.py is loaded from its own file with
the product's module under the import name it asks for (gaussian_renderer/__init__.py:16), so

  * `GaussianRasterizationSettings(image_height=..., ..., debug=...)` (:39-52, :142-155) constructs the
    product's 12-field NamedTuple from the reference's keywords,
  * `GaussianRasterizer(raster_settings=...)` (:54, :157) is the product's nn.Module,
  * `rasterizer(means3D=..., means2D=..., shs=..., colors_precomp=..., opacities=..., scales=..., rotations=...,
    cov3D_precomp=...)` (:103-111, :208-216) goes through the product's forward() -- its argument names, its
    "exactly one of" checks -- and arrives at ggsplat.rasterizer.rasterize_gaussians,

and ONLY that last function is replaced, by a recorder (no GPU in the authoring container; what the function
does with the nine positional arguments + settings is what the GPU tests cover).  What it receives must equal
render_args.npz -- the recording of the same scenarios against the stand-in.  A drift of forward()'s
signature, of a keyword name or of a NamedTuple field fails here.

Runs only where /root/reference exists (the reference never travels): `python tests/golden/check_reference_binding.py`,
and as tests/test_reference_binding.py in the CPU suite.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, "gaussian-garments_amd")

ARG_ORDER = ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")


class _CpuTorch:
    """The name `torch` inside the reference module: torch itself, minus the hard-coded device="cuda" of
    gaussian_renderer/__init__.py:29,132 (there is no GPU where this runs)."""

    def __getattr__(self, k):
        return getattr(torch, k)

    def zeros_like(self, *a, **kw):
        if str(kw.get("device", "")).startswith("cuda"):
            kw.pop("device")
        return torch.zeros_like(*a, **kw)


def load_reference_renderer():
    """gaussian_renderer/__init__.py of the reference as a module object, with the product's rasterizer module under the
    name it imports.  The two model classes it imports for type annotations only (:17-18) come as empty stand-ins (their
    files pull in the whole training stack); utils.sh_utils is the reference's own file.  sys.modules is left as found."""
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import diff_gaussian_rasterization_depth_alpha as product          # the REAL drop-in module
    assert os.path.realpath(product.__file__).startswith(os.path.realpath(PKG)), product.__file__
    saved = {k: sys.modules.get(k) for k in ("scene", "scene.mesh_gaussian_model", "scene.gaussian_model", "utils",
                                             "utils.sh_utils", "gaussian_renderer_ref")}
    try:
        scene = types.ModuleType("scene")
        scene.__path__ = []
        mgm = types.ModuleType("scene.mesh_gaussian_model")
        mgm.MeshGaussianModel = type("MeshGaussianModel", (), {})
        gm = types.ModuleType("scene.gaussian_model")
        gm.GaussianModel = type("GaussianModel", (), {})
        utils = types.ModuleType("utils")
        utils.__path__ = [os.path.join(REF, "utils")]
        sys.modules.update({"scene": scene, "scene.mesh_gaussian_model": mgm, "scene.gaussian_model": gm, "utils": utils})
        sys.modules.pop("utils.sh_utils", None)
        spec = importlib.util.spec_from_file_location("gaussian_renderer_ref", os.path.join(REF, "gaussian_renderer", "__init__.py"))
        GR = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(GR)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    # the reference's file bound the PRODUCT's classes
    assert GR.GaussianRasterizer is product.GaussianRasterizer
    assert GR.GaussianRasterizationSettings is product.GaussianRasterizationSettings
    GR.torch = _CpuTorch()
    return GR, product


def run(golden=os.path.join(HERE, "render_args.npz")):
    """Every scenario of render_args.npz through the reference's render() / doll_render() on the product module.
    Returns the number of (scenario, argument) comparisons made; raises AssertionError on the first difference."""
    NS = types.SimpleNamespace
    r = np.load(golden)
    GR, product = load_reference_renderer()
    import ggsplat.rasterizer as RZ
    calls = []

    def recorder(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        calls.append((settings, dict(zip(ARG_ORDER, (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                                    cov3Ds_precomp)))))
        n, H, W = means3D.shape[0], settings.image_height, settings.image_width
        return torch.zeros(3, H, W), torch.arange(n, dtype=torch.int32) % 3, torch.zeros(1, H, W), torch.zeros(1, H, W)
    real = (product.rasterize_gaussians, RZ.rasterize_gaussians)
    assert real[0] is real[1], "the drop-in module must call ggsplat.rasterizer.rasterize_gaussians"
    product.rasterize_gaussians = RZ.rasterize_gaussians = recorder
    n_checked = 0
    try:
        fix = {k[4:]: torch.tensor(r[k]) for k in r.files if k.startswith("fix_")}
        P = fix["_xyz"].shape[0]
        cam = NS(FoVx=float(r["cam_fov"][0]), FoVy=float(r["cam_fov"][1]), image_height=int(r["cam_size"][0]),
                 image_width=int(r["cam_size"][1]), world_view_transform=torch.tensor(r["cam_view"]),
                 full_proj_transform=torch.tensor(r["cam_proj"]), camera_center=torch.tensor(r["cam_center"]))
        mask, bg = torch.tensor(r["vis_mask"]), torch.tensor(r["bg"])

        def pc(with_shs=False, with_local=False):
            o = NS(_xyz=fix["_xyz"], active_sh_degree=1, max_sh_degree=1, get_xyz=fix["get_xyz"], get_opacity=fix["get_opacity"],
                   get_scaling=fix["get_scaling"], get_rotation=fix["get_rotation"], get_features=fix["get_features"],
                   get_covariance=lambda mod: torch.full((P, 6), float(mod)))
            if with_shs:
                o.shs = fix["shs"]
            if with_local:
                o.local_xyz, o.get_final_xyz = fix["local_xyz"], fix["get_final_xyz"]
            return o
        plain = dict(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        scenarios = {
            "default": (pc(), NS(**plain), {}),
            "s3": (pc(True, True), NS(**plain), dict(vis_mask=mask)),
            "python": (pc(), NS(debug=True, compute_cov3D_python=True, convert_SHs_python=True), dict(scaling_modifier=0.5)),
            "override": (pc(), NS(**plain), dict(override_color=fix["override"])),
            "override_masked": (pc(), NS(**plain), dict(override_color=fix["override"], vis_mask=mask)),
        }

        def check(name, rs, args):
            nonlocal n_checked
            # the settings object IS the product's NamedTuple, fields in the upstream order
            assert type(rs) is product.GaussianRasterizationSettings
            assert rs._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                                  "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
            got = np.array([rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree,
                            float(rs.prefiltered), float(rs.debug)], dtype=np.float64)
            assert np.array_equal(got, r[f"{name}_settings"]), (name, got, r[f"{name}_settings"])
            assert rs.bg is bg and rs.viewmatrix is cam.world_view_transform and rs.projmatrix is cam.full_proj_transform
            assert rs.campos is cam.camera_center
            assert sorted(k for k, v in args.items() if v is None) == [str(k) for k in r[f"{name}_none"]], name
            for k, v in args.items():
                if v is not None:
                    ref = r[f"{name}_arg_{k}"]
                    assert tuple(v.shape) == ref.shape, (name, k)
                    assert np.array_equal(v.detach().numpy(), ref), (name, k)      # same file, same torch: bit for bit
                    n_checked += 1
        for name, (model, pipe, kw) in scenarios.items():
            out = GR.render(cam, model, pipe, bg, **kw)
            rs, args = calls[-1]
            check(name, rs, args)
            assert bool(args["means2D"].requires_grad) == bool(r[f"{name}_means2D_requires_grad"])
            assert sorted(out) == [str(k) for k in r[f"{name}_out_keys"]]
            assert np.array_equal(out["visibility_filter"].numpy(), r[f"{name}_visibility"])
        doll = NS(xyz=fix["get_xyz"], opacity=fix["get_opacity"], scaling=fix["get_scaling"], rotation=fix["get_rotation"],
                  features=fix["get_features"], active_sh_degree=1, max_sh_degree=1, covariance=lambda mod: torch.full((P, 6), float(mod)))
        for name, kw in {"doll_default": {}, "doll_override_shs": dict(override_shs=fix["shs"]),
                         "doll_override_color": dict(override_color=fix["override"]),
                         "doll_masked": dict(override_shs=fix["shs"], vis_mask=mask)}.items():
            out = GR.doll_render(cam, doll, NS(**plain), bg, **kw)
            rs, args = calls[-1]
            check(name, rs, args)
            assert len(out) == int(r[f"{name}_n_outputs"]) and [list(o.shape) for o in out] == r[f"{name}_out_shapes"].tolist()
        assert len(calls) == 9
        # the product's forward() keeps upstream's "exactly one of" errors in front of the rasterizer (b1)
        rz = GR.GaussianRasterizer(raster_settings=calls[0][0])
        a = calls[0][1]
        for bad in (dict(shs=None), dict(colors_precomp=torch.zeros(P, 3)), dict(scales=None), dict(cov3D_precomp=torch.zeros(P, 6))):
            try:
                rz(**{**a, **bad})
            except Exception as e:
                assert "exactly one" in str(e) or "excatly one" in str(e), e
            else:
                raise AssertionError(f"no error for {sorted(bad)}")
        assert len(calls) == 9, "an invalid argument combination reached the rasterizer"
    finally:
        product.rasterize_gaussians, RZ.rasterize_gaussians = real
    return n_checked


if __name__ == "__main__":
    print("reference gaussian_renderer on the product module:", run(), "tensor arguments equal render_args.npz")
