"""HIP `fused.fused_bias_act` and `upfirdn2d.upfirdn2d` (drop-ins for the reference's StyleGAN2 extension
modules) against golden outputs of the reference's own PyTorch paths (fused_act.py CPU branch,
upfirdn2d_native), plus the derivative modes and the adjoint identity the reference's autograd wrappers rely on."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = np.load(os.path.join(os.path.dirname(__file__), "golden", "stylegan_ops.npz"))
SQRT2 = 2 ** 0.5


def test_fused_bias_act_forward_matches_reference():
    import fused
    x, b = torch.tensor(D["act_x"]).cuda(), torch.tensor(D["act_b"]).cuda()
    e = x.new_empty(0)
    y = fused.fused_bias_act(x, b, e, 3, 0, 0.2, SQRT2)
    assert np.allclose(y.cpu().numpy(), D["act_y_bias"], rtol=1e-6, atol=1e-7)
    y = fused.fused_bias_act(x, e, e, 3, 0, 0.2, SQRT2)
    assert np.allclose(y.cpu().numpy(), D["act_y_nobias"], rtol=1e-6, atol=1e-7)
    x2, b2 = torch.tensor(D["act2_x"]).cuda(), torch.tensor(D["act2_b"]).cuda()        # 2-D input: step_b = 1
    assert np.allclose(fused.fused_bias_act(x2, b2, e, 3, 0, 0.2, SQRT2).cpu().numpy(), D["act2_y"], rtol=1e-6, atol=1e-7)
    # linear act, half precision round trip
    yl = fused.fused_bias_act(x.half(), b.half(), e.half(), 1, 0, 0.2, 0.5)
    assert yl.dtype == torch.float16 and torch.allclose(yl.float(), ((x.half() + b.half().view(1, -1, 1, 1)).float() * 0.5), atol=2e-3)


def test_fused_bias_act_derivative_modes():
    """grad=1 (what FusedLeakyReLUFunctionBackward calls) equals autograd of the forward; grad=2 is zero."""
    import fused
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 4, 4, generator=g).cuda().requires_grad_(True)
    b = torch.randn(5, generator=g).cuda()
    ref = torch.nn.functional.leaky_relu(x + b.view(1, -1, 1, 1), 0.2) * SQRT2
    go = torch.randn(ref.shape, generator=g).cuda()
    (gx,) = torch.autograd.grad(ref, x, go)
    e = x.new_empty(0)
    out = fused.fused_bias_act(x.detach(), b, e, 3, 0, 0.2, SQRT2)
    gi = fused.fused_bias_act(go, e, out, 3, 1, 0.2, SQRT2)
    assert torch.allclose(gi, gx, rtol=1e-6, atol=1e-7)
    assert float(fused.fused_bias_act(go, e, out, 3, 2, 0.2, SQRT2).abs().max()) == 0.0


@pytest.mark.parametrize("name", ["blur", "blur3", "up2", "up2haar", "down2", "down2haar", "crop", "mixed"])
def test_upfirdn2d_matches_reference_native(name):
    import upfirdn2d as U
    inp = torch.tensor(D["ufd_in"]).cuda()                     # [2,3,9,11]
    k = torch.tensor(D[f"ufd_{name}_k"]).cuda()
    ux, uy, dx, dy, px0, px1, py0, py1 = [int(v) for v in D[f"ufd_{name}_cfg"]]
    B, Cn, Hh, Ww = inp.shape
    out = U.upfirdn2d(inp.reshape(-1, Hh, Ww, 1), k, ux, uy, dx, dy, px0, px1, py0, py1)
    ref = D[f"ufd_{name}_out"]
    assert tuple(out.shape) == (B * Cn, ref.shape[2], ref.shape[3], 1)
    assert np.allclose(out.view(B, Cn, ref.shape[2], ref.shape[3]).cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_upfirdn2d_adjoint_identity_and_minor():
    """The reference's backward is upfirdn2d(grad, flip(kernel), up<->down, g_pad) (upfirdn2d.py:128-141):
    check <upfirdn(x), y> == <x, backward(y)> with those parameters, and the minor (channels-last) axis."""
    import upfirdn2d as U
    g = torch.Generator().manual_seed(1)
    k = torch.randn(4, 3, generator=g).cuda()
    up, down, pad = (2, 1), (1, 2), (2, 1, 1, 2)
    x = torch.randn(5, 8, 6, 1, generator=g).cuda()
    out = U.upfirdn2d(x, k, up[0], up[1], down[0], down[1], *pad)
    y = torch.randn(out.shape, generator=g).cuda()
    kh, kw = k.shape
    in_h, in_w = 8, 6
    out_h, out_w = out.shape[1], out.shape[2]
    g_pad_x0, g_pad_y0 = kw - pad[0] - 1, kh - pad[2] - 1
    g_pad_x1 = in_w * up[0] - out_w * down[0] + pad[0] - up[0] + 1
    g_pad_y1 = in_h * up[1] - out_h * down[1] + pad[2] - up[1] + 1
    gx = U.upfirdn2d(y, torch.flip(k, [0, 1]), down[0], down[1], up[0], up[1], g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
    assert gx.shape == x.shape
    lhs, rhs = float((out.double() * y.double()).sum()), float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))
    xm = torch.randn(2, 7, 5, 3, generator=g).cuda()           # minor = 3 processed independently
    om = U.upfirdn2d(xm, k, 1, 1, 1, 1, 1, 1, 2, 0)
    for c in range(3):
        oc = U.upfirdn2d(xm[..., c:c + 1].contiguous(), k, 1, 1, 1, 1, 1, 1, 2, 0)
        assert torch.equal(om[..., c:c + 1], oc)


# ---- the StyleUNet shapes (minor = 1, major = batch x channels) through the tiled specialised kernels ---------------
def _k4():
    k = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k = k[None, :] * k[:, None]
    return k / k.sum()


def _haar():
    s = 1 / (2 ** 0.5)
    l, h = torch.tensor([[s, s]]), torch.tensor([[-s, s]])
    return {"ll": l.T * l, "lh": h.T * l, "hl": l.T * h, "hh": h.T * h}       # styleunet.py:371-384


# (name, kernel, up, down, pad) -- Blur of a stride-2 ConvLayer (pad (2,2)) and of an upsampling ModulatedConv2d (pad (1,1),
# kernel * 4), Upsample / Downsample (styleunet.py:32-71), Haar / inverse Haar (:387-425), and the backward forms the
# reference's autograd wrapper issues for them (up <-> down, flipped kernel, g_pad: upfirdn2d.py:128-141)
STYLE_CASES = [
    ("blur_pad22", _k4(), 1, 1, (2, 2, 2, 2)), ("blur_pad11_x4", _k4() * 4, 1, 1, (1, 1, 1, 1)), ("blur_pad21", _k4(), 1, 1, (2, 1, 2, 1)),
    ("upsample", _k4() * 4, 2, 1, (2, 1, 2, 1)), ("downsample", _k4(), 1, 2, (1, 1, 1, 1)),
    ("haar_ll", _haar()["ll"], 1, 2, (0, 0, 0, 0)), ("haar_hl", _haar()["hl"], 1, 2, (0, 0, 0, 0)),
    ("ihaar_lh", _haar()["lh"], 2, 1, (1, 0, 1, 0)), ("ihaar_hh", _haar()["hh"], 2, 1, (1, 0, 1, 0)),
    ("upsample_bwd", torch.flip(_k4() * 4, [0, 1]), 1, 2, (1, 2, 1, 2)), ("downsample_bwd", torch.flip(_k4(), [0, 1]), 2, 1, (2, 2, 2, 2)),
    ("crop_blur", _k4(), 1, 1, (-3, 2, 1, -5)),
]


@pytest.mark.parametrize("case", STYLE_CASES, ids=[c[0] for c in STYLE_CASES])
@pytest.mark.parametrize("shape", [(2, 8, 512, 512), (1, 3, 2048, 2048), (3, 5, 130, 77)], ids=["512", "2048", "ragged"])
def test_upfirdn2d_styleunet_shapes(case, shape):
    import upfirdn2d as U
    from oracle import stylegan_oracle as SO
    name, k, up, down, pad = case
    B, Cn, Hh, Ww = shape
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(B, Cn, Hh, Ww, generator=g).cuda()
    out = U.upfirdn2d(x.reshape(-1, Hh, Ww, 1), k.cuda(), up, up, down, down, *pad)
    ref = SO.upfirdn2d(x, k.cuda(), (up, up), (down, down), pad)
    assert tuple(out.shape) == (B * Cn, ref.shape[2], ref.shape[3], 1)
    err = float((out.view_as(ref) - ref).abs().max())
    assert err <= 2e-6 * max(1.0, float(ref.abs().max())), (name, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.float64, 1e-13)])
def test_stylegan_ops_native_dtypes(dtype, tol):
    """half and double are processed in place of being converted: output dtype = input dtype, double keeps double
    accuracy (a float round trip would leave ~1e-7), half matches a float computation rounded once."""
    import fused
    import upfirdn2d as U
    from oracle import stylegan_oracle as SO
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 96, 80, generator=g, dtype=torch.float64)
    b = torch.randn(6, generator=g, dtype=torch.float64)
    xd, bd = x.to(dtype).cuda(), b.to(dtype).cuda()
    e = xd.new_empty(0)
    # alpha and scale cross the ABI as C floats (upstream's op signature has `float alpha, float scale` too)
    a32, s32 = float(torch.tensor(0.2, dtype=torch.float32)), float(torch.tensor(SQRT2, dtype=torch.float32))
    y = fused.fused_bias_act(xd, bd, e, 3, 0, 0.2, SQRT2)
    ref = SO.fused_leaky_relu(xd.double(), bd.double(), a32, s32)
    assert y.dtype == dtype and float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max())
    gi = fused.fused_bias_act(xd, e, y, 3, 1, 0.2, SQRT2)            # derivative mode with a reference tensor
    refg = torch.where(ref > 0, xd.double(), xd.double() * a32) * s32
    assert gi.dtype == dtype and float((gi.double() - refg).abs().max()) <= tol * float(refg.abs().max())
    for name, k, up, down, pad in STYLE_CASES[:6] + [("generic", torch.randn(3, 2, generator=g), 1, 1, (1, 0, 2, 0))]:
        out = U.upfirdn2d(xd.reshape(-1, 96, 80, 1), k.to(dtype).cuda(), up, up, down, down, *pad)
        r = SO.upfirdn2d(xd.double(), k.to(dtype).double().cuda(), (up, up), (down, down), pad)
        assert out.dtype == dtype
        assert float((out.view_as(r).double() - r).abs().max()) <= tol * max(1.0, float(r.abs().max())), name


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.float32, 2e-6)], ids=["half", "float"])
@pytest.mark.parametrize("width,offset", [(80, 0), (82, 0), (81, 0), (80, 2), (80, 1), (84, 3)])
def test_upfirdn2d_vector_access_alignments(dtype, tol, width, offset):
    """The tiled kernels stage the patch and store the outputs 1, 2 or 4 elements per access, chosen from the row pitch, the
    plane size and the base alignment of each tensor: every combination (widths that are multiples of 4, of 2 only, odd;
    bases offset by 1, 2, 3 elements) gives the same values."""
    import upfirdn2d as U
    from oracle import stylegan_oracle as SO
    g = torch.Generator().manual_seed(width * 7 + offset)
    n = 3 * 70 * width
    flat = torch.randn(n + offset, generator=g).to(dtype).cuda()
    x = flat[offset:].view(3, 70, width, 1)
    assert x.data_ptr() % (4 * x.element_size()) == (offset * x.element_size()) % (4 * x.element_size())
    for name, k, up, down, pad in STYLE_CASES:
        out = U.upfirdn2d(x, k.to(dtype).cuda(), up, up, down, down, *pad)
        r = SO.upfirdn2d(x.view(1, 3, 70, width).double(), k.to(dtype).double().cuda(), (up, up), (down, down), pad)
        assert out.dtype == dtype
        assert float((out.view_as(r).double() - r).abs().max()) <= tol * max(1.0, float(r.abs().max())), (name, width, offset)


def test_stylegan_ops_reject_other_dtypes():
    import fused
    import upfirdn2d as U
    x = torch.zeros(1, 2, 4, 4, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="not implemented for"):
        fused.fused_bias_act(x, x.new_empty(0), x.new_empty(0), 3, 0, 0.2, 1.0)
    with pytest.raises(RuntimeError, match="not implemented for"):
        U.upfirdn2d(x.reshape(2, 4, 4, 1), torch.ones(2, 2, device="cuda"), 1, 1, 1, 1, 0, 0, 0, 0)
