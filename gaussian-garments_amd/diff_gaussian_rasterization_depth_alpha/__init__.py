"""Drop-in for the reference's external CUDA extension of the same import name.

The reference does (gaussian_renderer/__init__.py:16)
    from diff_gaussian_rasterization_depth_alpha import GaussianRasterizationSettings, GaussianRasterizer
and builds that module from an un-vendored CUDA repository (setup.sh:26-28).  Putting
THIS package on sys.path instead makes the reference's own render() / doll_render()
run on an MI355X through libggsplat.so (hand-written gfx950 HIP kernels), unchanged:
ROCm PyTorch reports the GPU as device type "cuda", so the reference's hard-coded
``device="cuda"`` strings keep working.

API (same names, argument meaning and error behaviour as the upstream module):
  GaussianRasterizationSettings  -- 12-field NamedTuple (call site :39-52)
  GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
        scales=None, rotations=None, cov3D_precomp=None) -> (color, radii, depth, alpha)   (call site :103-111)
  GaussianRasterizer.markVisible(positions) -> bool [P]
"""
from typing import NamedTuple

import torch
from torch import nn

from ggsplat.rasterizer import rasterize_gaussians

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of the upstream API: a point is visible iff its view-space z > 0.2."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix
            z = positions @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
