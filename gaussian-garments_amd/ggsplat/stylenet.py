"""A StyleGAN2-style U-Net on the two HIP ops of row f3 -- the SHAPE of the appearance network of stage 3, exercised as a
network.

The reference's appearance net (`scene/styleunet/styleunet.py:634-672`, built at `scene/avatar_net.py:21` with
inp_size = out_size = texture_size, 4 input channels, (sh_degree + 1)^2 * 3 + 3 output channels, style_dim = texture_size)
is PyTorch code whose only native parts are the extension modules `fused` and `upfirdn2d`; it runs unmodified on top of this
repo's drop-ins of those two modules.  The net itself is out of scope and is NOT reproduced here.
What this file holds is an independent, smaller network of the same KIND, written for this repo:

  * the two autograd ops every StyleGAN2 block is made of, each expressed -- forward, backward and double backward --
    through the ONE forward op of the extension module it stands on, the way `scene/styleunet/fused_act.py:33-130` and
    `scene/styleunet/upfirdn2d.py:98-184` do it: `bias_act` (fused.fused_bias_act) and `resample2d` (upfirdn2d.upfirdn2d);
  * plain-PyTorch twins of both (`impl="native"`), which the tests compare values and gradients against;
  * equalised convolutions, style-modulated convolutions with demodulation, blur-filtered resampling, Haar analysis /
    synthesis -- an encoder down to 8x8, a styled decoder back up with skip connections, at the reference's channel table
    (512 features up to 64x64, then 256 / 128 / 64 / 32 -- `styleunet.py:662-672` with channel_multiplier = 2).

`StyleUNetLite(size)` maps a [B, 4, size, size] condition texture + a [B, style_dim] style vector to a
[B, out_ch, size, size] texture.  `TexelOffsets` wraps it as the `net(gaussians, cam)` callable of
ggsplat.inner_step.appearance_step: texels are sampled at per-Gaussian UV coordinates into the position / SH offsets.
"""
from __future__ import annotations

import math
import os
from typing import Sequence, Tuple

# MIOpen's Find benchmarks every applicable solver the first time it sees a convolution shape; a trial kernel of its NHWC
# implicit-GEMM assembly family (backward data) faulted on the small-channel shapes of this network depending on the allocator's
# layout (profiles/r04_fault_triage.md).  Taken out of the trial list before the first convolution unless the caller decided
# otherwise (same default as tests/conftest.py and bench.py).
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")

import torch
import torch.nn as nn
import torch.nn.functional as F

CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32, 2048: 16}


# ------------------------------------------------------------------------------------------------------------------
# op 1: y = leaky_relu(x + bias[channel], slope) * gain, through fused.fused_bias_act
#   forward            : act = 3 (leaky ReLU), grad = 0
#   d/dx (a linear map of the incoming gradient, gated by the sign of the saved OUTPUT): act = 3, grad = 1, ref = y
#   second derivative  : the same gated map applied to the incoming second-order gradient (piecewise linear: no curvature)
# ------------------------------------------------------------------------------------------------------------------
def _sum_to_bias(g: torch.Tensor) -> torch.Tensor:
    return g.sum(dim=[0] + list(range(2, g.ndim)))


class _BiasActGrad(torch.autograd.Function):
    """g -> dL/dx = g * gain * (y > 0 ? 1 : slope)   (y = saved forward output)."""

    @staticmethod
    def forward(ctx, g, y, slope, gain):
        import fused
        ctx.save_for_backward(y)
        ctx.slope, ctx.gain = slope, gain
        empty = g.new_empty(0)
        return fused.fused_bias_act(g.contiguous(), empty, y, 3, 1, slope, gain)

    @staticmethod
    def backward(ctx, gg):
        import fused
        (y,) = ctx.saved_tensors
        empty = gg.new_empty(0)
        return fused.fused_bias_act(gg.contiguous(), empty, y, 3, 1, ctx.slope, ctx.gain), None, None, None


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, gain):
        import fused
        y = fused.fused_bias_act(x.contiguous(), bias if bias is not None else x.new_empty(0), x.new_empty(0), 3, 0, slope, gain)
        ctx.save_for_backward(y)
        ctx.slope, ctx.gain, ctx.has_bias = slope, gain, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gx = _BiasActGrad.apply(g, y, ctx.slope, ctx.gain)
        return gx, (_sum_to_bias(gx) if ctx.has_bias else None), None, None


def bias_act(x, bias=None, slope: float = 0.2, gain: float = math.sqrt(2.0), impl: str = "hip"):
    """leaky_relu(x + bias) * gain over [B, C, ...] (bias per channel)."""
    if impl == "native":
        if bias is not None:
            x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
        return F.leaky_relu(x, slope) * gain
    return _BiasAct.apply(x, bias, slope, gain)


# ------------------------------------------------------------------------------------------------------------------
# op 2: FIR resampling  y = decimate_down( pad( zero_insert_up(x) ) (*) k )  through upfirdn2d.upfirdn2d
#   The operator is linear in x; its adjoint is the same kind of operator with up <-> down, the kernel flipped, and the
#   padding that makes the sizes come out:  with n_in -> n_out = (n_in up + p0 + p1 - kw) / down + 1,
#       q0 = kw - 1 - p0,     q1 = n_in up - n_out down + p0 - up + 1
#   (derived by writing y[o] = sum_j k'[j] z[o down + j - p0], z[i up] = x[i], and collecting the coefficient of x[i]).
#   The adjoint's adjoint is the operator itself, so two Functions calling each other give every order of derivative.
# ------------------------------------------------------------------------------------------------------------------
def _fir(x, k, up, down, pad):
    import upfirdn2d as ext
    b, c, h, w = x.shape
    y = ext.upfirdn2d(x.reshape(b * c, h, w, 1), k, up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3])
    return y.reshape(b, c, y.shape[1], y.shape[2])


def _adjoint_pad(in_hw, out_hw, kshape, up, down, pad):
    (h, w), (oh, ow), (kh, kw) = in_hw, out_hw, kshape
    return (kw - 1 - pad[0], w * up[0] - ow * down[0] + pad[0] - up[0] + 1,
            kh - 1 - pad[2], h * up[1] - oh * down[1] + pad[2] - up[1] + 1)


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, up, down, pad):
        y = _fir(x, k, up, down, pad)
        ctx.k, ctx.up, ctx.down, ctx.pad = k, up, down, pad
        ctx.in_hw, ctx.out_hw = tuple(x.shape[2:]), tuple(y.shape[2:])
        return y

    @staticmethod
    def backward(ctx, g):
        qpad = _adjoint_pad(ctx.in_hw, ctx.out_hw, tuple(ctx.k.shape), ctx.up, ctx.down, ctx.pad)
        gx = _Resample.apply(g.contiguous(), torch.flip(ctx.k, [0, 1]), ctx.down, ctx.up, qpad)
        return gx, None, None, None, None


def resample2d_native(x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """The same operator in plain PyTorch ops (zero insertion, pad / crop, depth-wise correlation, decimation)."""
    b, c, h, w = x.shape
    z = x.new_zeros(b, c, h, up[1], w, up[0])
    z[:, :, :, 0, :, 0] = x
    z = z.reshape(b, c, h * up[1], w * up[0])
    z = F.pad(z, [max(pad[0], 0), max(pad[1], 0), max(pad[2], 0), max(pad[3], 0)])
    z = z[:, :, max(-pad[2], 0): z.shape[2] - max(-pad[3], 0), max(-pad[0], 0): z.shape[3] - max(-pad[1], 0)]
    kern = torch.flip(k.to(z.dtype), [0, 1])[None, None].expand(c, 1, -1, -1)
    return F.conv2d(z, kern, groups=c)[:, :, ::down[1], ::down[0]]


def resample2d(x, k, up: int = 1, down: int = 1, pad: Tuple[int, int] = (0, 0), impl: str = "hip"):
    """x [B,C,H,W]; k [kh,kw]; the same factors and padding in x and y."""
    upt, dnt, padt = (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1])
    if impl == "native":
        return resample2d_native(x, k, upt, dnt, padt)
    return _Resample.apply(x.contiguous(), k, upt, dnt, padt)


# ------------------------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------------------------
def binomial_kernel(taps: Sequence[float] = (1.0, 3.0, 3.0, 1.0)) -> torch.Tensor:
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


def haar_kernels():
    s = 1.0 / math.sqrt(2.0)
    lo, hi = torch.tensor([[s, s]]), torch.tensor([[-s, s]])
    return [lo.t() @ lo, hi.t() @ lo, lo.t() @ hi, hi.t() @ hi]         # LL, LH, HL, HH


class EqualConv(nn.Module):
    """Convolution with a runtime 1 / sqrt(fan_in) weight scale (equalised learning rate), optional bias + leaky ReLU."""

    def __init__(self, cin, cout, ksize, stride=1, act=True, impl="hip"):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, ksize, ksize))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.scale, self.stride, self.pad, self.act, self.impl = 1.0 / math.sqrt(cin * ksize * ksize), stride, ksize // 2, act, impl

    def forward(self, x, pad=None):
        y = F.conv2d(x, self.weight * self.scale, None if self.act else self.bias, stride=self.stride,
                     padding=self.pad if pad is None else pad)
        return bias_act(y, self.bias, impl=self.impl) if self.act else y


class Down(nn.Module):
    """Anti-aliased stride-2 convolution: binomial blur (pad so that the strided 3x3 sees the right border), conv stride 2."""

    def __init__(self, cin, cout, impl="hip"):
        super().__init__()
        self.register_buffer("k", binomial_kernel())
        self.conv = EqualConv(cin, cout, 3, stride=2, impl=impl)
        self.impl = impl

    def forward(self, x):
        return self.conv(resample2d(x, self.k, pad=(2, 2), impl=self.impl), pad=0)


class StyledConv(nn.Module):
    """Weight-modulated 3x3 convolution with demodulation (one grouped convolution for the batch), optional x2 up-sampling by
    transposed convolution + binomial blur, per-channel bias + leaky ReLU."""

    def __init__(self, cin, cout, style_dim, up=False, impl="hip"):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(1, cout, cin, 3, 3))
        self.affine = nn.Linear(style_dim, cin)
        nn.init.ones_(self.affine.bias)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.scale, self.up, self.impl, self.cin, self.cout = 1.0 / math.sqrt(cin * 9), up, impl, cin, cout
        if up:
            self.register_buffer("k", binomial_kernel() * 4.0)

    def forward(self, x, style):
        b, _, h, w = x.shape
        s = self.affine(style).view(b, 1, self.cin, 1, 1)
        wgt = self.weight * self.scale * s
        wgt = wgt * torch.rsqrt(wgt.pow(2).sum([2, 3, 4], keepdim=True) + 1e-8)
        x = x.reshape(1, b * self.cin, h, w)
        if self.up:
            wt = wgt.transpose(1, 2).reshape(b * self.cin, self.cout, 3, 3)
            y = F.conv_transpose2d(x, wt, stride=2, padding=0, groups=b)
            y = y.reshape(b, self.cout, y.shape[2], y.shape[3])
            y = resample2d(y, self.k, pad=(1, 1), impl=self.impl)
        else:
            y = F.conv2d(x, wgt.reshape(b * self.cout, self.cin, 3, 3), padding=1, groups=b)
            y = y.reshape(b, self.cout, h, w)
        return bias_act(y, self.bias, impl=self.impl)


class ToTexture(nn.Module):
    """1x1 projection to the output channels (in the Haar domain: 4 sub-bands per channel); the running output of the coarser
    level is brought up by Haar synthesis -> binomial x2 up-sampling -> Haar analysis and added."""

    def __init__(self, cin, out_ch, impl="hip"):
        super().__init__()
        self.proj = EqualConv(cin, out_ch * 4, 1, act=False, impl=impl)
        self.register_buffer("k", binomial_kernel() * 4.0)
        for i, hk in enumerate(haar_kernels()):
            self.register_buffer(f"h{i}", hk)
        self.impl = impl

    def haar_up(self, t):                                 # [B, 4C, h, w] -> [B, C, 2h, 2w]
        ll, lh, hl, hh = t.chunk(4, 1)
        ks = [self.h0, -self.h1, -self.h2, self.h3]
        return sum(resample2d(part, k, up=2, pad=(1, 0), impl=self.impl) for part, k in zip((ll, lh, hl, hh), ks))

    def haar_down(self, t):                               # [B, C, 2h, 2w] -> [B, 4C, h, w]
        return torch.cat([resample2d(t, k, down=2, impl=self.impl) for k in (self.h0, self.h1, self.h2, self.h3)], 1)

    def forward(self, feat, prev=None):
        out = self.proj(feat)
        if prev is not None:
            full = resample2d(self.haar_up(prev), self.k, up=2, pad=(2, 1), impl=self.impl)
            out = out + self.haar_down(full)
        return out


class StyleUNetLite(nn.Module):
    """Encoder: conv -> (Down)* to 8x8.  Decoder: (StyledConv up, StyledConv) per level with the encoder feature of the same
    resolution merged in by a 3x3 conv, texture accumulated in the Haar domain, synthesised once at the end."""

    def __init__(self, size: int = 512, in_ch: int = 4, out_ch: int = 51, style_dim: int = 512, middle: int = 8,
                 impl: str = "hip", channels=None):
        super().__init__()
        ch = dict(CHANNELS if channels is None else channels)
        self.size, self.style_dim, self.impl = size, style_dim, impl
        half = size // 2
        self.mapping = nn.Sequential(nn.Linear(style_dim, style_dim), nn.LeakyReLU(0.2), nn.Linear(style_dim, style_dim))
        self.stem = EqualConv(in_ch, ch[half], 3, impl=impl)
        self.stem_down = Down(ch[half], ch[half], impl=impl)
        self.downs, self.merges = nn.ModuleList(), nn.ModuleList()
        res, cin = half, ch[half]
        self.enc_res = [res]
        while res > middle:
            res //= 2
            self.downs.append(Down(cin, ch[res], impl=impl))
            cin = ch[res]
            self.enc_res.append(res)
        self.ups, self.convs, self.heads = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        while res < half:
            res *= 2
            self.ups.append(StyledConv(cin, ch[res], style_dim, up=True, impl=impl))
            self.merges.append(EqualConv(2 * ch[res], ch[res], 3, impl=impl))
            self.convs.append(StyledConv(ch[res], ch[res], style_dim, impl=impl))
            self.heads.append(ToTexture(ch[res], out_ch, impl=impl))
            cin = ch[res]

    def forward(self, cond, style):
        w = self.mapping(F.normalize(style, dim=1) * math.sqrt(style.shape[1]))
        x = self.stem_down(self.stem(cond))
        skips = [x]
        for d in self.downs:
            x = d(x)
            skips.append(x)
        tex = None
        for i, (up, merge, conv, head) in enumerate(zip(self.ups, self.merges, self.convs, self.heads)):
            x = up(x, w)
            x = merge(torch.cat([x, skips[-2 - i]], 1))
            x = conv(x, w)
            tex = head(x, tex)
        return self.heads[-1].haar_up(tex)                 # [B, out_ch, size, size]


class TexelOffsets(nn.Module):
    """`net(gaussians, cam) -> (xyz_offset [P,3], sh_offset [P,K,3], vis_mask)` for ggsplat.inner_step.appearance_step: the
    texture predicted by StyleUNetLite from a fixed condition map and a camera-dependent style vector, read at the
    Gaussians' UV coordinates (bilinear), split into 3 position + 3 K SH channels (scene/avatar_net.py:58-87 has this role)."""

    def __init__(self, net: StyleUNetLite, uv: torch.Tensor, sh_k: int, vis_mask: torch.Tensor, cond: torch.Tensor,
                 xyz_gain: float = 0.002, sh_gain: float = 0.03):
        super().__init__()
        self.net, self.k = net, sh_k
        self.register_buffer("grid", (uv * 2.0 - 1.0).reshape(1, 1, -1, 2))
        self.register_buffer("vis", vis_mask)
        self.register_buffer("cond", cond)
        self.pose = nn.Linear(12, net.style_dim)
        self.xyz_gain, self.sh_gain = xyz_gain, sh_gain

    def forward(self, gaussians, cam):
        view = cam.world_view_transform.reshape(-1)[:12].reshape(1, 12)
        tex = self.net(self.cond, self.pose(view))
        feat = F.grid_sample(tex, self.grid, mode="bilinear", align_corners=False)[0, :, 0].t()      # [P, 3 + 3K]
        return feat[:, :3] * self.xyz_gain, feat[:, 3:].reshape(-1, self.k, 3) * self.sh_gain, self.vis
