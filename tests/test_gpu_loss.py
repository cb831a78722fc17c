"""Fused photometric loss (ggs_photometric_forward / _backward) against the PyTorch restatement of
utils/loss_utils.py (oracle/host_oracle.py, itself pinned to golden vectors of the reference):
values and gradients, with / without mask, ragged sizes, batched views, unequal upstream weights."""
import os
import time

import numpy as np
import pytest
import torch

from helpers import rel_l1
from oracle import host_oracle as HO

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _ref(img, gt, mask, lam, wa=1.0, wb=1.0):
    x = img.clone().requires_grad_(True)
    l_img = HO.l1_loss(x, gt, mask) * (1.0 - lam)
    l_ssim = 1.0 - HO.ssim(x, gt, mask) * lam
    (wa * l_img + wb * l_ssim).backward()
    return float(l_img.detach()), float(l_ssim.detach()), x.grad


# the kernels stream 64-column strips in bands of 34 rows, four bands per workgroup: sizes on, one over and one under those
# edges, one row / one column images, and a size with a partly idle last workgroup (137 rows = 5 bands)
@pytest.mark.parametrize("H,W,use_mask", [(48, 64, True), (48, 64, False), (37, 53, True), (70, 33, False), (11, 9, True),
                                          (34, 64, True), (35, 65, False), (33, 63, True), (137, 130, True), (1, 200, False),
                                          (150, 1, True)])
def test_fused_loss_matches_reference_restatement(H, W, use_mask):
    from ggsplat.loss import fused_photometric_loss
    g = torch.Generator().manual_seed(H * 100 + W)
    img, gt = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    mask = (torch.rand(1, H, W, generator=g) > 0.3).float() if use_mask else None
    lam = 0.2
    r_img, r_ssim, r_grad = _ref(img, gt, mask, lam, 1.0, 0.7)
    x = img.clone().cuda().requires_grad_(True)
    l_img, l_ssim = fused_photometric_loss(x, gt.cuda(), None if mask is None else mask.cuda(), lam)
    (l_img + 0.7 * l_ssim).backward()
    assert abs(float(l_img.detach()) - r_img) < 2e-6 and abs(float(l_ssim.detach()) - r_ssim) < 2e-6
    assert rel_l1(x.grad, r_grad) <= 1e-4


def test_fused_loss_on_reference_golden():
    """Directly against the numbers the reference's own l1_loss / ssim produced (tests/golden/loss.npz)."""
    from ggsplat.loss import fused_photometric_loss
    d = np.load(os.path.join(G, "loss.npz"))
    a, b, m = torch.tensor(d["img1"]).cuda(), torch.tensor(d["img2"]).cuda(), torch.tensor(d["mask"]).cuda()
    for tag, mask in (("nomask", None), ("mask", m)):
        x = a.clone().requires_grad_(True)
        l_img, l_ssim = fused_photometric_loss(x, b, mask, 1.0)          # lambda = 1 isolates SSIM: l_ssim = 1 - ssim
        assert abs((1.0 - float(l_ssim.detach())) - float(d[f"ssim_{tag}"])) < 2e-6
        (1.0 - l_ssim).backward()
        assert rel_l1(x.grad, d[f"ssim_grad_{tag}"]) <= 1e-4
        x = a.clone().requires_grad_(True)
        l_img, l_ssim = fused_photometric_loss(x, b, mask, 0.0)          # lambda = 0 isolates L1
        assert abs(float(l_img.detach()) - float(d[f"l1_{tag}"])) < 1e-6
        l_img.backward()
        assert rel_l1(x.grad, d[f"l1_grad_{tag}"]) <= 1e-6


def test_fused_loss_batched_views_and_1080p_speed():
    from ggsplat.loss import fused_photometric_loss, l1_loss, ssim
    g = torch.Generator().manual_seed(1)
    V, H, W = 3, 1080, 1920
    img, gt = torch.rand(V, 3, H, W, generator=g).cuda(), torch.rand(V, 3, H, W, generator=g).cuda()
    mask = (torch.rand(V, 1, H, W, generator=g) > 0.2).float().cuda()
    x = img.clone().requires_grad_(True)
    l_img, l_ssim = fused_photometric_loss(x, gt, mask, 0.2)
    (l_img + l_ssim).sum().backward()
    for v in range(V):                                       # per-view PyTorch composition on the GPU
        y = img[v].clone().requires_grad_(True)
        a = l1_loss(y, gt[v], mask[v]) * 0.8
        b = 1.0 - ssim(y + 0, gt[v].clone(), mask[v]) * 0.2
        (a + b).backward()
        assert abs(float(a.detach()) - float(l_img[v].detach())) < 2e-6 and abs(float(b.detach()) - float(l_ssim[v].detach())) < 2e-6
        assert rel_l1(x.grad[v], y.grad) <= 1e-4
    # timing (informational, printed with -s): fused vs PyTorch composition, fwd+bwd per 1080p view
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n
    def fused():
        z = img[:1].clone().requires_grad_(True)
        a, b = fused_photometric_loss(z, gt[:1], mask[:1], 0.2)
        (a + b).sum().backward()
    def torch_ops():
        z = img[0].clone().requires_grad_(True)
        (l1_loss(z, gt[0], mask[0]) * 0.8 + 1.0 - ssim(z + 0, gt[0].clone(), mask[0]) * 0.2).backward()
    # best of three trials each: a single host hiccup (the box also runs the profiler's daemons) must not fail a parity suite
    tf, tt = min(timed(fused) for _ in range(3)), min(timed(torch_ops) for _ in range(3))
    print(f"\\nphotometric loss fwd+bwd @1080p: fused {tf*1e3:.3f} ms, PyTorch ops {tt*1e3:.3f} ms ({tt/tf:.1f}x)")
    assert tf < tt


@pytest.mark.parametrize("V,H,W,use_mask,density", [(1, 200, 330, True, 0.15), (2, 137, 130, False, 0.3), (1, 360, 640, True, 0.05),
                                                   (3, 70, 33, True, 0.5), (1, 96, 128, False, 0.0), (1, 96, 128, True, 1.0)])
def test_region_of_interest_form(V, H, W, use_mask, density):
    """ggs_photometric_*_roi with the tile list lengths of a forward: the loss values are those of the plain form; dL/dimage is
    the plain form's, bit for bit, on every pixel of a non-empty tile (all the rasterizer's backward ever reads) -- in fact in
    every 64 x 12 box that overlaps one -- and the boxes that overlap none are not written at all."""
    import ctypes as C
    from ggsplat._lib import check, lib, ptr
    L = lib()
    dev = "cuda"
    g = torch.Generator().manual_seed(V * 1000 + H + W)
    img, gt = torch.rand(V, 3, H, W, generator=g).to(dev), torch.rand(V, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(V, 1, H, W, generator=g) > 0.3).float().to(dev) if use_mask else None
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tc = ((torch.rand(V, gy * gx, generator=g) < density).int() * 7).to(dev)
    w = torch.tensor([[0.8, -0.2]] * V, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(tile_count):
        sums = torch.empty(V, 2, device=dev)
        scratch = torch.empty(L.ggs_photometric_scratch_bytes(V, H, W), device=dev, dtype=torch.uint8)
        check(L.ggs_photometric_forward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tile_count), ptr(sums),
                                            ptr(scratch), stream), "fwd")
        d = torch.full((V, 3, H, W), 123.0, device=dev)
        check(L.ggs_photometric_backward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tile_count), ptr(scratch),
                                             ptr(w), ptr(d), stream), "bwd")
        return sums, d

    s_full, d_full = run(None)
    s_roi, d_roi = run(tc)
    assert float((s_full - s_roi).abs().max()) <= 1e-5 * float(s_full.abs().max())          # (atomic summation order)
    assert not (d_full == 123.0).any()
    # pixel masks: inside a non-empty tile; inside a box (64 columns x 12 rows) that overlaps a non-empty tile
    tile_px = (tc.reshape(V, gy, gx) != 0).repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W]
    box = torch.zeros(V, H, W, dtype=torch.bool, device=dev)
    for y0 in range(0, H, 12):
        for x0 in range(0, W, 64):
            hit = tile_px[:, y0:y0 + 12, x0:x0 + 64].reshape(V, -1).any(1)
            box[:, y0:y0 + 12, x0:x0 + 64] = hit[:, None, None]
    for ch in range(3):
        a, b = d_roi[:, ch], d_full[:, ch]
        assert torch.equal(a[box], b[box])                          # computed: identical to the plain form
        assert (a[~box] == 123.0).all()                             # not computed: not written
    assert bool((box | ~tile_px).all())                             # every pixel of a non-empty tile lies in an active box
    if density == 0.0:
        assert not box.any()
    if density == 1.0:
        assert box.all()


def _blob_mask(V, H, W, g, kind):
    """silhouette-like masks: a disc per view ("disc"), nothing ("empty"), everything ("full"), salt-and-pepper ("noise")"""
    if kind == "empty":
        return torch.zeros(V, 1, H, W)
    if kind == "full":
        return torch.ones(V, 1, H, W)
    if kind == "noise":
        return (torch.rand(V, 1, H, W, generator=g) > 0.3).float()
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    m = torch.zeros(V, 1, H, W)
    for v in range(V):
        cx, cy = W * (0.3 + 0.4 * float(torch.rand(1, generator=g))), H * (0.3 + 0.4 * float(torch.rand(1, generator=g)))
        r = 0.18 * min(H, W)
        m[v, 0] = (((xx - cx) ** 2 + (yy - cy) ** 2) < r * r).float()
    return m


@pytest.mark.parametrize("V,H,W", [(1, 200, 330), (2, 137, 130), (3, 70, 33), (1, 16, 16), (4, 97, 260)])
def test_mask_tile_occupancy(V, H, W):
    """ggs_mask_tiles: non-zero mask pixels per 16x16 tile, ragged edges included."""
    from ggsplat.loss import mask_tile_occupancy
    g = torch.Generator().manual_seed(V + H + W)
    m = (torch.rand(V, 1, H, W, generator=g) > 0.6).float() * torch.rand(V, 1, H, W, generator=g)
    m[:, :, : H // 2, : W // 3] = 0
    got = mask_tile_occupancy(m.cuda())
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(V, gy * 16, gx * 16)
    pad[:, :H, :W] = (m[:, 0] != 0).float()
    want = pad.reshape(V, gy, 16, gx, 16).sum(dim=(2, 4)).reshape(V, gy * gx).int()
    assert got.dtype is torch.int32 and torch.equal(got.cpu(), want)


@pytest.mark.parametrize("V,H,W,kind,density", [(1, 360, 640, "disc", 0.05), (1, 200, 330, "disc", 0.15), (2, 137, 130, "disc", 0.1),
                                                (3, 170, 233, "disc", 0.1), (1, 96, 128, "empty", 0.1), (1, 96, 128, "full", 0.2),
                                                (1, 150, 200, "noise", 0.1), (4, 70, 33, "disc", 0.5), (1, 1080, 1920, "disc", 0.08)])
def test_sparse_mask_form(V, H, W, kind, density):
    """ggs_photometric_forward_sparse: the sums of the region-of-interest form to summation order, and -- behind the unchanged
    backward pass -- its dL/dimage bit for bit on every box that pass writes (tiles with a list both inside and OUTSIDE the
    mask: the maps of a masked-out box are still produced when the backward will read them)."""
    import ctypes as C
    from ggsplat._lib import check, lib, ptr
    from ggsplat.loss import mask_tile_occupancy
    L = lib()
    dev = "cuda"
    g = torch.Generator().manual_seed(V * 1000 + H + W)
    img, gt = torch.rand(V, 3, H, W, generator=g).to(dev), torch.rand(V, 3, H, W, generator=g).to(dev)
    mask = _blob_mask(V, H, W, g, kind).to(dev)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # lists where the mask is (the garment) plus a few anywhere (Gaussians that left the silhouette)
    in_mask = mask_tile_occupancy(mask).cpu() != 0
    tc = ((in_mask & (torch.rand(V, gy * gx, generator=g) < 0.8)) | (torch.rand(V, gy * gx, generator=g) < density * 0.2)).int() * 5
    tc = tc.to(dev)
    mt = mask_tile_occupancy(mask)
    w = torch.tensor([[0.8, -0.2]] * V, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(sparse, table=False):
        sums = torch.empty(V, 2, device=dev)
        scratch = torch.full((L.ggs_photometric_scratch_bytes(V, H, W),), 0xFF, device=dev, dtype=torch.uint8)   # NaN maps
        if sparse and table:
            tabs = torch.tensor([mt[v].data_ptr() for v in range(V)], dtype=torch.int64, device=dev)
            check(L.ggs_photometric_forward_sparse(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), None, ptr(tabs),
                                                   ptr(sums), ptr(scratch), stream), "fwd sparse (table)")
        elif sparse:
            check(L.ggs_photometric_forward_sparse(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(mt), None,
                                                   ptr(sums), ptr(scratch), stream), "fwd sparse")
        else:
            check(L.ggs_photometric_forward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(sums),
                                                ptr(scratch), stream), "fwd roi")
        d = torch.full((V, 3, H, W), 123.0, device=dev)
        check(L.ggs_photometric_backward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(scratch),
                                             ptr(w), ptr(d), stream), "bwd")
        return sums, d

    s_roi, d_roi = run(False)
    for table in (False, True):
        s_sp, d_sp = run(True, table)
        assert float((s_roi - s_sp).abs().max()) <= 2e-6 * float(s_roi.abs().max()), (s_roi, s_sp)
        assert torch.equal(d_roi, d_sp)                              # incl. the 123s of the boxes nobody writes
    assert torch.isfinite(d_roi).all()
    # and against the PyTorch restatement of the reference's loss
    from ggsplat.loss import l1_loss, ssim
    for v in range(V):
        want_l1 = float(l1_loss(img[v], gt[v], mask[v])) * 3 * H * W
        want_ss = float(ssim(img[v].clone(), gt[v].clone(), mask[v])) * 3 * H * W
        assert abs(float(s_sp[v, 0]) - want_l1) <= 1e-4 * max(want_l1, 1.0)
        assert abs(float(s_sp[v, 1]) - want_ss) <= 2e-5 * want_ss


def test_sparse_mask_form_through_the_python_wrapper_and_its_errors():
    from ggsplat._lib import GgsError, check, lib, ptr
    from ggsplat.loss import fused_photometric_loss, mask_tile_occupancy
    g = torch.Generator().manual_seed(11)
    H, W = 120, 200
    img = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g).cuda()
    mask = _blob_mask(1, H, W, g, "disc")[0].cuda()
    mt = mask_tile_occupancy(mask)
    tc = (mt.reshape(1, -1) != 0).int()
    outs = []
    for tiles in (None, mt):
        img.grad = None
        a, b = fused_photometric_loss(img, gt, mask, 0.2, tile_count=tc, mask_tiles=tiles)
        (a + b).backward()
        outs.append((float(a.detach()), float(b.detach()), img.grad.clone()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * abs(outs[0][0]) and abs(outs[0][1] - outs[1][1]) <= 1e-6 * abs(outs[0][1])
    assert torch.equal(outs[0][2], outs[1][2])
    with pytest.raises(ValueError, match="mask_tiles needs"):
        fused_photometric_loss(img, gt, mask, 0.2, tile_count=None, mask_tiles=mt)
    with pytest.raises(ValueError, match="mask_tiles needs"):
        fused_photometric_loss(img, gt, None, 0.2, tile_count=tc, mask_tiles=mt)
    with pytest.raises(RuntimeError, match="GPU only"):
        mask_tile_occupancy(mask.cpu())
    # the C entry point refuses the combinations the wrapper refuses, and a missing table
    L = lib()
    sums = torch.empty(1, 2, device="cuda")
    scratch = torch.empty(L.ggs_photometric_scratch_bytes(1, H, W), device="cuda", dtype=torch.uint8)
    x = img.detach()
    assert L.ggs_photometric_forward_sparse(1, H, W, ptr(x), ptr(gt), ptr(mask), None, None, None, ptr(mt), None, ptr(sums),
                                            ptr(scratch), None) != 0
    assert L.ggs_photometric_forward_sparse(1, H, W, ptr(x), ptr(gt), None, None, None, ptr(tc), ptr(mt), None, ptr(sums),
                                            ptr(scratch), None) != 0
    assert L.ggs_photometric_forward_sparse(1, H, W, ptr(x), ptr(gt), ptr(mask), None, None, ptr(tc), None, None, ptr(sums),
                                            ptr(scratch), None) != 0
    assert L.ggs_mask_tiles(1, H, W, None, ptr(mt), None) != 0 and L.ggs_mask_tiles(0, H, W, ptr(mask), ptr(mt), None) != 0


def test_steps_with_the_region_of_interest_loss_give_the_same_gradients():
    """registration_step(fused_loss=True) now takes the loss gradient only where the render backward reads it: every parameter
    gradient equals the one obtained with the full dL/dimage (tile_count=None)."""
    from types import SimpleNamespace
    import ggsplat.inner_step as IS
    from ggsplat import synthetic as S
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    W, H = 320, 200
    v, f = S.skirt_mesh(30, 40)
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=0)
    cam = S.rig_cameras(n_rings=1, n_az=4, width=W, height=H, f=260.0)[1]
    gtr = torch.Generator().manual_seed(3)
    gt = torch.rand(3, H, W, generator=gtr).cuda()
    mask = (torch.rand(1, H, W, generator=gtr) > 0.2).float().cuda()
    bg = torch.zeros(3, device="cuda")
    grads = []
    for roi in (True, False):
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        m.training_setup(IS.DEFAULT_OPT, is_ff=True, optimizer="torch")
        orig = IS.R.last_tile_count
        if not roi:
            IS.R.last_tile_count = lambda: None
        try:
            IS.registration_step(m, cam, gt, mask, bg, optimizer_step=False, fused_loss=True)
        finally:
            IS.R.last_tile_count = orig
        grads.append({n: p.grad.clone() for n, p in zip(["mesh.v", "_xyz", "_f_dc", "_f_rest", "_opacity", "_scaling", "_rotation"],
                                                        m.parameters()) if p.grad is not None})
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) >= 5
    for k in grads[0]:
        assert rel_l1(grads[0][k], grads[1][k]) <= 1e-6, k
