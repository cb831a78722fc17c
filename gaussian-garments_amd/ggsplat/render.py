"""render() / doll_render() -- behaviour-identical host-side mirror of
gaussian_renderer/__init__.py:21-122 and :124-221 of the reference, on top of the HIP rasterizer.

Same signature, same choice logic (cov3D python path, SH python path, override colour, vis_mask
gather, pc.shs / get_final_xyz selection for the s3 model), same return dict.  With the drop-in
module `diff_gaussian_rasterization_depth_alpha` of this repo on sys.path the reference's own
render() works unmodified as well; this mirror exists so the inner-step harness and the tests do
not need the reference tree (it never travels to the GPU box).
"""
import math
from types import SimpleNamespace

import torch

from diff_gaussian_rasterization_depth_alpha import GaussianRasterizationSettings, rasterize_gaussians

from .sh import eval_sh


def _settings(cam, pc_sh_degree, pipe, bg_color, scaling_modifier):
    tf = getattr(cam, "tanfov", None)
    if tf is not None:      # static camera of a captured step: the tangents live in a device tensor [1,2]
        return SimpleNamespace(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfov=tf, tanfovx=None, tanfovy=None,
            bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=pc_sh_degree, campos=cam.camera_center, prefiltered=False,
            debug=pipe.debug)
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc_sh_degree, campos=cam.camera_center, prefiltered=False, debug=pipe.debug)


def _sel(t, mask):
    return t[mask] if (t is not None and mask is not None) else t


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mode=None, vis_mask=None):
    """Render one view.  bg_color must live on the GPU."""
    # zero tensor that carries the gradient of the 2-D (screen-space) means back to the caller.  The reference builds it as
    # `zeros(requires_grad=True) + 0` with retain_grad(); a leaf gets the same `.grad` through AccumulateGrad without the
    # add node and the hook (~12 us of host time per call in a loop whose GPU work is 0.1 ms).
    screenspace_points = torch.zeros_like(pc._xyz, dtype=pc._xyz.dtype, requires_grad=True, device=pc._xyz.device)
    settings = _settings(viewpoint_camera, pc.active_sh_degree, pipe, bg_color, scaling_modifier)
    use_final = getattr(pc, "local_xyz", None) is not None
    means3D = pc.get_final_xyz if use_final else pc.get_xyz
    try:
        means3D.retain_grad()
    except Exception:
        pass
    means2D = screenspace_points
    opacity = pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.shs if getattr(pc, "shs", None) is not None else pc.get_features
    else:
        colors_precomp = override_color

    mask_radii = None
    if vis_mask is not None and getattr(pipe, "mask_by_opacity", False):
        # Same image and gradients as the gather below (a zero-opacity splat is culled before binning), but every shape
        # stays static, which a captured hipGraph needs -- the boolean gather has a data-dependent length.
        opacity = opacity * vis_mask.to(opacity.dtype).reshape(-1, 1)
        mask_radii, vis_mask = vis_mask, None
    if vis_mask is not None:                    # only render visible Gaussians (s3)
        means3D, means2D, shs = _sel(means3D, vis_mask), _sel(means2D, vis_mask), _sel(shs, vis_mask)
        colors_precomp, opacity = _sel(colors_precomp, vis_mask), _sel(opacity, vis_mask)
        scales, rotations = _sel(scales, vis_mask), _sel(rotations, vis_mask)
        cov3D_precomp = _sel(cov3D_precomp, vis_mask)

    # (GaussianRasterizer(settings)(...) of the reference = this call behind an nn.Module construction and __call__)
    rendered_image, radii, depth, alpha = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacity, scales, rotations,
                                                              cov3D_precomp, settings)

    if mask_radii is not None:
        radii = radii * mask_radii.to(radii.dtype)
    # "tile_count" (not a key of the reference's dict): list lengths of the 16x16 tiles of THIS forward, taken right behind the
    # rasterizer call so that a later forward on the same thread cannot be mistaken for it; the region-of-interest form of the
    # fused loss reads it (ggsplat.loss.fused_photometric_loss(..., tile_count=...)).
    from . import rasterizer as _R
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "3dposition": means3D, "depth": depth, "alpha": alpha, "tile_count": _R.last_tile_count()}


def doll_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, override_shs=None,
                vis_mask=None):
    """Forward-only variant used by inference.py: attribute names xyz / opacity / scaling / rotation / features,
    returns (image, depth, alpha)."""
    screenspace_points = torch.zeros_like(pc.xyz, dtype=pc.xyz.dtype, requires_grad=True, device=pc.xyz.device)
    settings = _settings(viewpoint_camera, pc.active_sh_degree, pipe, bg_color, scaling_modifier)
    means3D, means2D, opacity = pc.xyz, screenspace_points, pc.opacity
    scales, rotations = pc.scaling, pc.rotation
    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    else:
        shs = override_shs if override_shs is not None else pc.features
    if vis_mask is not None:
        means3D, means2D, shs = _sel(means3D, vis_mask), _sel(means2D, vis_mask), _sel(shs, vis_mask)
        colors_precomp, opacity = _sel(colors_precomp, vis_mask), _sel(opacity, vis_mask)
        scales, rotations = _sel(scales, vis_mask), _sel(rotations, vis_mask)
    image, radii, depth, alpha = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacity, scales, rotations, None,
                                                     settings)
    return image, depth, alpha
