"""Photometric loss of the inner steps -- mirror of utils/loss_utils.py:17-68 (masked L1, 11x11
Gaussian-window SSIM via grouped conv2d).  Stays PyTorch-ROCm in this round (SURVEY section 8f ranks the
fused HIP loss kernel as "next" #1).  Like the reference, ssim() multiplies img1 / img2 by the mask
IN PLACE (loss_utils.py:44-46): callers hand it the rasterizer's output tensor, which is why the
rasterizer never saves its outputs for backward."""
from math import exp

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt, mask=None):
    if mask is None:
        return torch.abs(network_output - gt).mean()
    return torch.abs((network_output - gt) * mask).mean()


def _window(window_size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)


def ssim(img1, img2, mask=None, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    if mask is not None:
        img1 *= mask
        img2 *= mask
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)
