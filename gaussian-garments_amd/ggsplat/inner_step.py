"""Inner optimisation steps of stage 2 (registration) and stage 3 (appearance) -- host-side counterparts
of s2_registration.py:238-327 and s3_appearance.py:115-147, on the HIP rasterizer and the fused mesh
binding.  Same order of operations and the same loss composition; what is NOT here (out of scope,
SURVEY section 2): cloth energies, densify/prune, the StyleUNet (a caller-supplied `net` stands in for it),
data loading, logging."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .loss import fused_photometric_loss, l1_loss, ssim
from .render import render

DEFAULT_OPT = SimpleNamespace(                      # arguments/__init__.py:74-116 (the values the steps read)
    position_lr_init=0.00016, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
    percent_dense=0.01, lambda_dssim=0.2, lambda_xyz=1e-2, threshold_xyz=1.0, lambda_scale=1.0,
    threshold_scale=0.6, threshold_opacity=0.75, lambda_opacity=0.01, only_foreground_loss=True)
DEFAULT_PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)


def _photometric(image, gt_image, m, lam, fused: bool):
    """The two image terms of both loops: L1 * (1 - lambda) and 1 - SSIM * lambda.  fused=True: one pair of HIP
    kernels (ggs_photometric_*); fused=False: the reference's PyTorch composition (loss.py), which also masks
    `image` and `gt_image` in place like the reference does."""
    if fused:
        return fused_photometric_loss(image, gt_image, m, lam)
    return l1_loss(image, gt_image, m) * (1.0 - lam), 1.0 - ssim(image, gt_image, m) * lam


def registration_step(gaussians, viewpoint_cam, gt_image, mask, bg, opt=DEFAULT_OPT, pipe=DEFAULT_PIPE,
                      first_frame_template: bool = True, track_densification: bool = True,
                      optimizer_step: bool = True, fused_loss: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of the s2 loop: update_face_coor -> render -> L1 (1 - lambda) + (1 - ssim lambda)
    [+ xyz / scale hinges on the first template frame] -> backward -> densification stats -> Adam step."""
    gaussians.update_face_coor()
    pkg = render(viewpoint_cam, gaussians, pipe, bg)
    image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
    m = mask if opt.only_foreground_loss else None
    l_img, l_ssim = _photometric(image, gt_image, m, opt.lambda_dssim, fused_loss)
    loss_dict = {"img": l_img, "ssim": l_ssim}
    if first_frame_template:
        loss_dict["xyz"] = F.relu(gaussians._xyz[vis].norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz
        loss_dict["scale"] = F.relu(gaussians.scaling_activation(gaussians._scaling[vis]) - opt.threshold_scale
                                    ).norm(dim=1).mean() * opt.lambda_scale
    loss = sum(loss_dict.values())
    loss.backward()
    with torch.no_grad():
        if first_frame_template and track_densification:
            gaussians.max_radii2D[vis] = torch.max(gaussians.max_radii2D[vis], radii[vis].to(gaussians.max_radii2D.dtype))
            gaussians.add_densification_stats(vsp, vis)
        if optimizer_step and gaussians.optimizer is not None:
            gaussians.optimizer.step()
            gaussians.optimizer.zero_grad()
    loss_dict["loss"] = loss.detach()
    loss_dict["render_pkg"] = pkg
    return loss_dict


def appearance_step(gaussians, net: Callable, viewpoint_cam, gt_image, mask, bg, optimizer=None,
                    opt=DEFAULT_OPT, pipe=DEFAULT_PIPE, fused_loss: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of the s3 loop.  `net(gaussians, cam) -> (xyz_offset [P,3], sh_offset [P,K,3], vis_mask [P])`
    stands in for AvatarNet.forward (scene/avatar_net.py:58-87): it sets local_xyz = _xyz + offset and
    shs = get_features + offset, then render(..., vis_mask=vis_mask) and the five-term loss."""
    gaussians.update_face_coor()
    xyz_off, sh_off, vis_mask = net(gaussians, viewpoint_cam)
    gaussians.local_xyz = gaussians._xyz + xyz_off
    gaussians.shs = gaussians.get_features + sh_off
    pkg = render(viewpoint_cam, gaussians, pipe, bg, vis_mask=vis_mask)
    image = pkg["render"]
    m = mask if opt.only_foreground_loss else None
    l_img, l_ssim = _photometric(image, gt_image, m, opt.lambda_dssim, fused_loss)
    loss_dict = {"img": l_img, "ssim": l_ssim,
                 "xyz": F.relu(gaussians.local_xyz.norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz,
                 "scale": F.relu(gaussians.scaling_activation(gaussians._scaling) - opt.threshold_scale
                                 ).norm(dim=1).mean() * opt.lambda_scale,
                 "opacity": F.relu(opt.threshold_opacity - gaussians.get_opacity).mean() * opt.lambda_opacity}
    loss = sum(loss_dict.values())
    loss.backward()
    if optimizer is not None:
        with torch.no_grad():
            optimizer.step()
            optimizer.zero_grad()
    loss_dict["loss"] = loss.detach()
    loss_dict["render_pkg"] = pkg
    return loss_dict
