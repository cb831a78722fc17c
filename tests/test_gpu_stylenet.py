"""The two StyleGAN2 ops of row f3 exercised AS A NETWORK: ggsplat.stylenet expresses the autograd of
`fused.fused_bias_act` and `upfirdn2d.upfirdn2d` -- forward, backward, double backward -- through the same two forward ops,
the way scene/styleunet/fused_act.py:33-130 and scene/styleunet/upfirdn2d.py:98-184 do, and stacks StyleGAN2 blocks on them
at the channel table of scene/styleunet/styleunet.py:662-672.  Everything is compared with the same modules on their
plain-PyTorch (`impl="native"`) paths: same weights, same inputs, values and every parameter gradient."""
import copy
import time

import pytest
import torch

from ggsplat import stylenet as SN

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().sum() / (b.double().abs().sum() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_bias_act_all_orders(dtype):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 7, 9, 11, generator=g, dtype=dtype).cuda().requires_grad_(True)
    b = torch.randn(7, generator=g, dtype=dtype).cuda().requires_grad_(True)
    w = torch.randn(3, 7, 9, 11, generator=g, dtype=dtype).cuda()
    v = torch.randn(3, 7, 9, 11, generator=g, dtype=dtype).cuda()
    res = []
    for impl in ("hip", "native"):
        y = SN.bias_act(x, b, impl=impl)
        gx, gb = torch.autograd.grad((y * w).sum(), (x, b), create_graph=True)
        (ggw,) = torch.autograd.grad((gx * v).sum(), (w,), allow_unused=True) if w.requires_grad else (None,)
        # second order: d/dx of <gx, v> is zero almost everywhere (piecewise linear); what is non-trivial is the dependence of
        # gx on the incoming gradient, i.e. the backward of the backward: check it through a differentiable upstream gradient
        up = w.clone().requires_grad_(True)
        gx2 = torch.autograd.grad(SN.bias_act(x, b, impl=impl), x, up, create_graph=True)[0]
        (gup,) = torch.autograd.grad((gx2 * v).sum(), up)
        res.append((y.detach(), gx.detach(), gb.detach(), gup.detach()))
    # (alpha and the gain cross the C ABI as `float`, like upstream's op: sqrt(2) carries fp32 rounding in the double path too)
    tol = 1e-6 if dtype is torch.float32 else 1e-7
    for a, r in zip(*res):
        assert _rel(a, r) <= tol


CASES = [("blur", SN.binomial_kernel(), 1, 1, (2, 1)), ("up", SN.binomial_kernel() * 4, 2, 1, (2, 1)),
         ("down", SN.binomial_kernel(), 1, 2, (2, 2)), ("haar", SN.haar_kernels()[1], 1, 2, (0, 0)),
         ("ihaar", SN.haar_kernels()[2], 2, 1, (1, 0)), ("blur_up_tail", SN.binomial_kernel() * 4, 1, 1, (1, 1))]


@pytest.mark.parametrize("name,k,up,down,pad", CASES, ids=[c[0] for c in CASES])
def test_resample2d_all_orders(name, k, up, down, pad):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 18, 22, generator=g).cuda().requires_grad_(True)
    k = k.cuda()
    res = []
    for impl in ("hip", "native"):
        y = SN.resample2d(x, k, up, down, pad, impl=impl)
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda().requires_grad_(True)
        (gx,) = torch.autograd.grad((y * w).sum(), x, create_graph=True)
        v = torch.randn(x.shape, generator=torch.Generator().manual_seed(4)).cuda()
        (gw,) = torch.autograd.grad((gx * v).sum(), w)          # backward of the backward = the forward operator applied to v
        res.append((y.detach(), gx.detach(), gw.detach()))
    for a, r in zip(*res):
        assert a.shape == r.shape and _rel(a, r) <= 2e-6


def _pair(size, out_ch, style_dim, channels=None, seed=5):
    torch.manual_seed(seed)
    hip = SN.StyleUNetLite(size=size, in_ch=4, out_ch=out_ch, style_dim=style_dim, impl="hip", channels=channels).cuda()
    nat = SN.StyleUNetLite(size=size, in_ch=4, out_ch=out_ch, style_dim=style_dim, impl="native", channels=channels).cuda()
    nat.load_state_dict(copy.deepcopy(hip.state_dict()))
    return hip, nat


def _fwd_bwd(net, cond, style, w):
    for p in net.parameters():
        p.grad = None
    out = net(cond, style)
    (out * w).sum().backward()
    return out.detach(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_small_network_values_and_every_gradient(dtype):
    """Same weights, same inputs, HIP ops against native ops.  In double every parameter gradient agrees to the 1e-7 of the
    fp32 gain constant that crosses the C ABI; in single the texture agrees to 1e-5 and the gradients to 5e-3 -- a leaky-ReLU
    unit whose pre-activation is within rounding of zero takes the other slope in one of the two paths, a discrete change that
    the deepest parameters (the mapping layers feed every modulated convolution) see as ~1e-3; no such flip survives in double."""
    ch = {k: min(v, 24) for k, v in SN.CHANNELS.items()}
    hip, nat = _pair(64, 7, 32, ch)
    hip, nat = hip.to(dtype), nat.to(dtype)
    g = torch.Generator().manual_seed(6)
    cond, style = torch.randn(2, 4, 64, 64, generator=g).cuda().to(dtype), torch.randn(2, 32, generator=g).cuda().to(dtype)
    w = torch.randn(2, 7, 64, 64, generator=g).cuda().to(dtype)
    oh, gh = _fwd_bwd(hip, cond, style, w)
    on, gn = _fwd_bwd(nat, cond, style, w)
    assert oh.shape == (2, 7, 64, 64) and _rel(oh, on) <= (1e-6 if dtype is torch.float64 else 1e-5)
    assert set(gh) == set(gn) and len(gh) >= 30
    tol = 1e-6 if dtype is torch.float64 else 5e-3
    for n in gn:
        assert _rel(gh[n], gn[n]) <= tol, n


def _record_preactivation_signs(monkeypatch_target, store):
    """Wrap ggsplat.stylenet.bias_act so that every call appends (impl, pre-activation) of its leaky ReLU."""
    orig = SN.bias_act

    def spy(x, bias=None, slope=0.2, gain=2.0 ** 0.5, impl="hip"):
        pre = x.detach() if bias is None else x.detach() + bias.detach().reshape(1, -1, *([1] * (x.ndim - 2)))
        store.setdefault(impl, []).append(pre)
        return orig(x, bias, slope, gain, impl)
    monkeypatch_target.setattr(SN, "bias_act", spy)


def _count_flips(pre):
    """(units, units whose pre-activation sign differs between the two paths, largest |pre-activation| among those relative
    to its layer's r.m.s.)"""
    units = flips = 0
    worst_rel = 0.0
    for a, b in zip(pre["hip"], pre["native"]):
        assert a.shape == b.shape
        units += a.numel()
        d = (a > 0) != (b > 0)
        n = int(d.sum())
        flips += n
        if n:
            rms = float(b.double().pow(2).mean().sqrt())
            worst_rel = max(worst_rel, float(torch.maximum(a.abs(), b.abs())[d].max()) / rms)
    return units, flips, worst_rel


def test_fp32_gradient_gap_is_explained_by_slope_flips(monkeypatch):
    """The fp32 bars of this file (5e-3 for the small net, 1e-2 at texture 512) are 50-100 x looser than the path's 1e-4, and
    the stated reason is discrete: a leaky-ReLU unit whose pre-activation is within rounding of zero takes the other slope in
    one of the two paths (HIP ops / native ops) and the deepest parameters see the whole difference.  Checked here instead of
    asserted in prose: (1) the two paths disagree on the SIGN of a pre-activation for a handful of units in ~10^6, (2) every one
    of them sits within 1e-5 of its layer's r.m.s. of zero (rounding distance), (3) the texture itself agrees to 1e-5 and, in
    double -- where no unit flips -- every gradient agrees to 1e-6 (test_small_network_values_and_every_gradient[float64])."""
    ch = {k: min(v, 24) for k, v in SN.CHANNELS.items()}
    hip, nat = _pair(64, 7, 32, ch)
    g = torch.Generator().manual_seed(6)
    cond, style = torch.randn(2, 4, 64, 64, generator=g).cuda(), torch.randn(2, 32, generator=g).cuda()
    w = torch.randn(2, 7, 64, 64, generator=g).cuda()
    pre = {}
    _record_preactivation_signs(monkeypatch, pre)
    oh, gh = _fwd_bwd(hip, cond, style, w)
    on, gn = _fwd_bwd(nat, cond, style, w)
    assert len(pre["hip"]) == len(pre["native"]) >= 10
    units, flips, worst_rel = _count_flips(pre)
    frac = flips / units
    worst = max(_rel(gh[n], gn[n]) for n in gn)
    print(f"\n[slope flips] {flips} of {units} leaky-ReLU units ({frac:.2e}) take different slopes in the two paths; the largest "
          f"|pre-activation| among them is {worst_rel:.1e} of its layer's r.m.s.; worst parameter-gradient difference {worst:.1e}, "
          f"texture {_rel(oh, on):.1e}")
    assert units > 400_000 and frac <= 1e-4              # a handful of units ...
    assert flips == 0 or worst_rel <= 1e-4               # ... each within rounding of zero
    assert worst <= (5e-3 if flips else 1e-4)            # no flip -> the path's own bar; with flips the discrete bar


def test_reference_channel_table_texture_512(monkeypatch):
    """Texture size 512 (the reference's default, s3_appearance.py:61), 4 -> 51 channels ((3 + 1)^2 * 3 + 3, avatar_net.py:21),
    style_dim 512, channel table of styleunet.py:662-672: forward + backward on the HIP ops against the native paths, timed.
    The fp32 gradient bar is 1e-4 (the path's own) unless leaky-ReLU units take different slopes in the two paths; then it is
    1e-2, the flipped units are counted, and each must sit within rounding of zero."""
    hip, nat = _pair(512, 51, 512)
    g = torch.Generator().manual_seed(7)
    cond, style = torch.randn(1, 4, 512, 512, generator=g).cuda(), torch.randn(1, 512, generator=g).cuda()
    w = torch.randn(1, 51, 512, 512, generator=g).cuda()
    pre = {}
    with monkeypatch.context() as mp:
        _record_preactivation_signs(mp, pre)
        oh, gh = _fwd_bwd(hip, cond, style, w)
        on, gn = _fwd_bwd(nat, cond, style, w)
    assert oh.shape == (1, 51, 512, 512) and _rel(oh, on) <= 1e-4
    units, flips, worst_rel = _count_flips(pre)
    worst = max(_rel(gh[n], gn[n]) for n in gn)
    print(f"\n[StyleUNetLite 512] {flips} of {units} leaky-ReLU units take different slopes in the two paths (largest |pre-activation| "
          f"{worst_rel:.1e} of the layer r.m.s.); worst parameter-gradient difference {worst:.1e}")
    assert flips <= 1e-5 * units and (flips == 0 or worst_rel <= 1e-4)
    assert worst <= (1e-2 if flips else 1e-4), worst
    times = {}
    for name, net in (("hip", hip), ("native", nat)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _fwd_bwd(net, cond, style, w)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 3
    n_par = sum(p.numel() for p in hip.parameters())
    print(f"\n[StyleUNetLite 512, {n_par / 1e6:.1f} M parameters] fwd+bwd: HIP ops {times['hip'] * 1e3:.1f} ms, native ops "
          f"{times['native'] * 1e3:.1f} ms; worst parameter-gradient rel. L1 {worst:.2e}")
