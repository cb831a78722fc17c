"""Photometric loss of the inner steps.  Two forms of the same arithmetic:
  * l1_loss / ssim: the PyTorch mirror of utils/loss_utils.py:17-68 (masked L1, 11x11 Gaussian-window SSIM via grouped
    conv2d), reference semantics included: ssim() multiplies img1 / img2 by the mask IN PLACE (loss_utils.py:44-46) --
    callers hand it the rasterizer's output tensor, which is why the rasterizer never saves its outputs for backward;
  * fused_photometric_loss: the fused HIP kernels of SURVEY section 8 row f1 (csrc/ggs_loss.hip, ggs_photometric_*): both
    loss terms and dL/dimage in two row-streaming passes, optionally only where the render backward reads the gradient
    (tile_count = the region-of-interest form)."""
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


# ---------------------------------------------------------------------------------------------------
# Fused HIP version (libggsplat.so: ggs_photometric_forward / _backward) -- the two loss terms of the inner
# steps and the gradient w.r.t. the rendered image in two tile passes instead of ~25 PyTorch kernels.
def _f32c4(t):
    """Contiguous float32 [V, C, H, W] view of an image tensor; the common case (already that) costs attribute reads only."""
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    return t if t.dim() == 4 else t.unsqueeze(0)


_scratch_bytes = {}          # (V, H, W) -> bytes of the inter-pass scratch (one C call per shape instead of one per loss)
_loss_consts = {}            # (device, lambda, n) -> the constant vectors of the value / gradient epilogues


def _consts(dev, lam, n):
    k = (dev.index, lam, n)
    c = _loss_consts.get(k)
    if c is None:
        if len(_loss_consts) > 64:
            _loss_consts.clear()
        c = _loss_consts[k] = (torch.tensor([[(1.0 - lam) / n, -lam / n]], device=dev), torch.tensor([[0.0, 1.0]], device=dev),
                               torch.tensor([[1.0 - lam, -lam]], device=dev))
    return c


class _FusedPhotometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, mask, lambda_dssim, tile_count=None, mask_tiles=None):
        import ctypes as C
        from ._lib import check, lib, ptr, stream_ptr
        if image.device.type != "cuda":
            raise RuntimeError("ggsplat fused loss runs on the GPU only (no CPU path in the product)")
        img, g = _f32c4(image), _f32c4(gt)
        V, _, H, W = img.shape
        m = None
        if mask is not None:
            m = _f32c4(mask.reshape(-1, 1, H, W))
            m = m if m.shape[0] == V else m.expand(V, 1, H, W).contiguous()
        L = lib()
        dev = img.device
        sums = torch.empty(V, 2, device=dev, dtype=torch.float32)
        nbytes = _scratch_bytes.get((V, H, W))
        if nbytes is None:
            nbytes = _scratch_bytes[(V, H, W)] = int(L.ggs_photometric_scratch_bytes(V, H, W))
        scratch = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        tc = None
        if tile_count is not None:
            tc = tile_count.contiguous()
            if tc.dtype not in (torch.int32, torch.uint32) or tc.numel() != V * ((H + 15) // 16) * ((W + 15) // 16):
                raise ValueError("fused_photometric_loss: tile_count must be int32 [V, ceil(H/16) * ceil(W/16)]")
        if mask_tiles is not None:
            if m is None or tc is None:
                raise ValueError("fused_photometric_loss: mask_tiles needs the mask and tile_count (the sparse-mask form skips "
                                 "only the boxes whose derivative maps the region-of-interest backward does not read)")
            mt = mask_tiles.contiguous()
            if mt.dtype not in (torch.int32, torch.uint32) or mt.device != dev:
                raise ValueError("fused_photometric_loss: mask_tiles must be the int32 device tensor mask_tile_occupancy() returns")
            if mt.numel() == tc.numel() // V and V > 1:          # one mask for all views, like `mask`
                mt = mt.reshape(1, -1).expand(V, -1).contiguous()
            if mt.numel() != tc.numel():
                raise ValueError("fused_photometric_loss: mask_tiles must hold ceil(H/16) * ceil(W/16) counts per view")
            check(L.ggs_photometric_forward_sparse(V, H, W, ptr(img), ptr(g), ptr(m), None, None, ptr(tc), ptr(mt), None,
                                                   ptr(sums), ptr(scratch), stream_ptr(dev)),
                  "ggs_photometric_forward_sparse")
        else:
            check(L.ggs_photometric_forward_roi(V, H, W, ptr(img), ptr(g), ptr(m), None, None, ptr(tc), ptr(sums), ptr(scratch),
                                                stream_ptr(dev)),
                  "ggs_photometric_forward")
        lam = float(lambda_dssim)
        scale, bias, _ = _consts(dev, lam, 3.0 * H * W)
        # The backward re-reads the image.  It is NOT copied (24 MB per 1080p view): like the rasterizer's autograd node this
        # one records the version counter of the caller's tensor and refuses a backward after an in-place edit -- what
        # save_for_backward would do, without keeping the producer's graph alive through a saved output.
        shares = img.data_ptr() == image.data_ptr()
        ctx.saved = (img, g, m, scratch, lam, image.shape, tc, image if shares else None, image._version if shares else 0)
        out = torch.addcmul(bias, sums, scale)          # [V, 2]: (sum|x - y| (1 - lambda) / n,  1 - sum ssim lambda / n)
        return out[:, 0], out[:, 1]

    @staticmethod
    def backward(ctx, g_img, g_ssim):
        import ctypes as C
        from ._lib import check, lib, ptr, stream_ptr
        img, g, m, scratch, lam, in_shape, tc, src, ver = ctx.saved
        if src is not None and src._version != ver:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                               f"the image handed to fused_photometric_loss is at version {src._version}; expected version {ver} "
                               "(clone it first if it must be edited between the loss and its backward)")
        V, _, H, W = img.shape
        dev = img.device
        _, _, wvec = _consts(dev, lam, 3.0 * H * W)
        if g_img is None or g_ssim is None:
            z = torch.zeros(V, device=dev)
            g_img, g_ssim = (z if g_img is None else g_img), (z if g_ssim is None else g_ssim)
        w = torch.stack((g_img.reshape(V), g_ssim.reshape(V)), dim=1).float() * wvec
        # region of interest: the kernels leave dL/dimage alone outside the boxes that touch a non-empty tile -- zero there, so
        # that a gradient summed with other consumers of the image stays finite
        dimg = torch.empty_like(img) if tc is None else torch.zeros_like(img)
        check(lib().ggs_photometric_backward_roi(V, H, W, ptr(img), ptr(g), ptr(m), None, None, ptr(tc), ptr(scratch), ptr(w),
                                                 ptr(dimg), stream_ptr(dev)),
              "ggs_photometric_backward")
        return dimg.reshape(in_shape), None, None, None, None, None


def mask_tile_occupancy(mask):
    """int32 [V, ceil(H/16) * ceil(W/16)]: the number of non-zero pixels of each mask ([V,1,H,W], [1,H,W] or [H,W], on the GPU) per
    16x16 tile -- the table the sparse-mask form of the fused loss takes (`mask_tiles=`).  Compute it ONCE per mask (the masks
    of the loops are per-camera constants, s2_registration.py:246-258)."""
    from ._lib import check, lib, ptr, stream_ptr
    if mask.device.type != "cuda":
        raise RuntimeError("ggsplat.mask_tile_occupancy runs on the GPU only (no CPU path in the product)")
    H, W = mask.shape[-2:]
    m = mask.detach().reshape(-1, H, W)
    m = m if m.dtype is torch.float32 else m.float()
    m = m if m.is_contiguous() else m.contiguous()
    V = m.shape[0]
    out = torch.empty(V, ((H + 15) // 16) * ((W + 15) // 16), device=m.device, dtype=torch.int32)
    check(lib().ggs_mask_tiles(V, H, W, ptr(m), ptr(out), stream_ptr(m.device)), "ggs_mask_tiles")
    return out


def fused_photometric_loss(image, gt, mask=None, lambda_dssim: float = 0.2, tile_count=None, mask_tiles=None):
    """(l1_loss(image, gt, mask) * (1 - lambda), 1 - ssim(image, gt, mask) * lambda), per view when the inputs
    are batched [V,3,H,W], through the fused HIP kernels.  Unlike the reference's ssim() it does not mask
    `image` / `gt` in place (the masking happens inside the kernels).  The image is NOT copied for the backward: an in-place
    edit of it between this call and loss.backward() raises autograd's "modified by an inplace operation" error.
    tile_count (rasterizer.last_tile_count() of the forward that rendered `image`): region-of-interest form -- the loss
    VALUES are the same, the gradient w.r.t. the image is computed only where the rasterizer's backward reads it (pixels of
    tiles that have a list) and is zero elsewhere: right for every parameter behind the rasterizer, not a full dL/dimage.
    mask_tiles (mask_tile_occupancy(mask), with tile_count): sparse-mask form -- the forward pass skips the boxes that hold no
    mask pixel; same values up to fp32 summation order, for silhouette masks (on a dense mask it is slower than without)."""
    l_img, l_ssim = _FusedPhotometric.apply(image, gt, mask, lambda_dssim, tile_count, mask_tiles)
    if image.dim() == 3:
        return l_img[0], l_ssim[0]
    return l_img, l_ssim
