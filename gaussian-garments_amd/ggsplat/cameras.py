"""Camera matrices consumed by the rasterizer -- host-side mirror of the reference.

Mirrors (same names, argument meaning, layouts):
  * ``utils/graphics_utils.py:38-49``  getWorld2View2
  * ``utils/graphics_utils.py:51-75``  getProjectionMatrix (off-centre principal point)
  * ``utils/graphics_utils.py:77-81``  fov2focal / focal2fov
  * ``scene/cameras.py:17-62``         Camera  (world_view_transform = W2C^T,
                                        full_proj_transform = (P W2C)^T, camera_center)

The reference hard-codes ``.cuda()`` in ``Camera.__init__``; here the device is a
parameter (default "cuda", which is what ROCm PyTorch calls an MI355X) so that the
same class also feeds the CPU oracle in tests.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """R is the camera-to-world rotation (the reference stores extrinsic[:, :3]^T,
    scene/dataset_readers.py:143), t the world-to-camera translation.  fp64 math,
    cast to fp32 at the end like the reference."""
    w2c = np.zeros((4, 4))
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = np.asarray(t)
    w2c[3, 3] = 1.0
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(c2w))


def getProjectionMatrix(znear, zfar, fovX, fovY, fx, fy, cx, cy, w, h):
    """Perspective matrix with an off-centre principal point; pixel = fx x/z + cx - 0.5."""
    top, bottom = cy / fy * znear, -(h - cy) / fy * znear
    right, left = cx / fx * znear, -(w - cx) / fx * znear
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


class Camera(torch.nn.Module):
    """Attribute-compatible with ``scene/cameras.py:Camera`` for everything ``render()`` reads."""

    def __init__(self, colmap_id=None, R=None, T=None, FoVx=None, FoVy=None, fx=None, fy=None, cx=None, cy=None,
                 image=None, gt_alpha_mask=None, image_name=None, uid=None,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, data_device="cuda",
                 image_width=None, image_height=None):
        super().__init__()
        self.uid, self.colmap_id, self.image_name = uid, colmap_id, image_name
        self.R, self.T = R, T
        self.FoVx, self.FoVy = FoVx, FoVy
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.data_device = torch.device(data_device)
        if image is not None:
            self.original_image = image.clamp(0.0, 1.0).to(self.data_device)
            self.image_width = self.original_image.shape[2]
            self.image_height = self.original_image.shape[1]
        else:                                   # synthetic cameras carry no ground-truth image
            self.original_image = None
            self.image_width, self.image_height = int(image_width), int(image_height)
        self.gt_alpha_mask = gt_alpha_mask
        self.zfar, self.znear = 100.0, 0.01
        self.trans, self.scale = trans, scale
        dev = self.data_device
        self.world_view_transform = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1).to(dev)
        self.projection_matrix = getProjectionMatrix(
            znear=self.znear, zfar=self.zfar, fovX=FoVx, fovY=FoVy, fx=fx, fy=fy, cx=cx, cy=cy,
            w=self.image_width, h=self.image_height).transpose(0, 1).to(dev)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3]


def look_at_camera(eye, target, *, width, height, fx, fy, cx, cy, up=(0.0, 1.0, 0.0), uid=None, device="cuda") -> Camera:
    """Synthetic camera in the reference's convention (+z forward, +y down, +x right)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    x = np.cross(-up, f)
    x /= np.linalg.norm(x)
    y = np.cross(f, x)
    R = np.stack([x, y, f], axis=1)             # camera-to-world rotation (columns = camera axes)
    T = -R.T @ eye
    return Camera(R=R, T=T, FoVx=focal2fov(fx, width), FoVy=focal2fov(fy, height), fx=fx, fy=fy, cx=cx, cy=cy,
                  image_width=width, image_height=height, uid=uid, data_device=device)
