"""GPU visibility ray cast (ggs_visibility) against an exhaustive float64 Moller-Trumbore on the host:
first-hit triangle ids and the visible mask, on a closed tube seen from outside (about half the Gaussians are
occluded by the near wall), with barycentric anchors, and the brute-force fallback when the grid lists overflow.
open3d / Embree (what the reference calls) is not installable here: this row is 'parity unpinned' against it."""
import numpy as np
import pytest
import torch

from ggsplat import synthetic as S

pytestmark = pytest.mark.gpu


def _first_hit_ref(v, f, cam, tg):
    v, tg, cam = v.double().numpy(), tg.double().numpy(), cam.double().numpy()
    f = f.numpy()
    d = tg - cam
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    best_t = np.full(len(tg), np.inf)
    best_f = np.full(len(tg), -1)
    margin = np.full(len(tg), np.inf)          # gap between the two nearest hits: rays with a tiny gap are ambiguous in fp32
    for i in range(len(tg)):
        pv = np.cross(d[i], e2)
        det = (e1 * pv).sum(1)
        ok = det != 0
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        tv = cam - v0
        u = (tv * pv).sum(1) * inv
        qv = np.cross(tv, e1)
        w = (qv * d[i]).sum(1) * inv
        t = (e2 * qv).sum(1) * inv
        hit = ok & (u >= 0) & (u <= 1) & (w >= 0) & (u + w <= 1) & (t > 0)
        if hit.any():
            ts = np.where(hit, t, np.inf)
            o = np.argsort(ts)
            best_t[i], best_f[i] = ts[o[0]], o[0]
            # how close the ray passes to the edge of the winning triangle / to the runner-up
            edge = min(u[o[0]], w[o[0]], 1 - u[o[0]] - w[o[0]])
            margin[i] = min(edge, (ts[o[1]] - ts[o[0]]) if np.isfinite(ts[o[1]]) else np.inf)
    return best_f, margin


@pytest.mark.parametrize("with_bary", [False, True])
def test_first_hit_and_mask(with_bary):
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f = S.skirt_mesh(40, 24, jitter=2e-3, seed=2)              # 1920 faces, closed around, open top / bottom
    p = S.skirt_gaussian_params(f.shape[0], 0, seed=2)
    bc = None
    if with_bary:
        g = torch.Generator().manual_seed(0)
        b = torch.rand(f.shape[0], 3, generator=g) + 0.05
        bc = b / b.sum(1, keepdim=True)
    m = MeshGaussianModel.from_tensors(v, f, p, 0, device="cuda", gs_bc=bc)
    cam = torch.tensor([2.0, 1.3, 0.7])
    mask, first = m.get_visible_mask(cam.cuda(), return_first_hit=True)
    tg = m.get_anchor_points().cpu()
    ref_f, margin = _first_hit_ref(v, f, cam, tg)
    sure = margin > 1e-5                                           # not grazing an edge / a tie in fp32
    assert sure.mean() > 0.98
    assert np.array_equal(first.cpu().numpy()[sure], ref_f[sure])
    ref_mask = ref_f == p["binding"].numpy()
    assert np.array_equal(mask.cpu().numpy()[sure], ref_mask[sure])
    assert 0.25 < ref_mask.mean() < 0.75                           # near wall visible, far wall hidden


def test_overflow_falls_back_to_exhaustive_search():
    from ggsplat._lib import check, lib, ptr
    import ctypes as C
    v, f = S.skirt_mesh(24, 12, jitter=2e-3, seed=4)
    dev = "cuda"
    vv, ff = v.cuda().contiguous(), f.cuda().contiguous()
    tg = vv[ff].mean(1).contiguous()
    bd = torch.arange(f.shape[0], device=dev)
    cam = torch.tensor([0.2, 3.0, 2.0], device=dev)
    L = lib()
    outs = []
    for cap in (32 * f.shape[0] + 65536, 8):                       # 8 ids: guaranteed overflow -> exhaustive kernel
        scratch = torch.empty(L.ggs_visibility_scratch_bytes(f.shape[0], v.shape[0], cap), device=dev, dtype=torch.uint8)
        mask = torch.empty(f.shape[0], device=dev, dtype=torch.uint8)
        first = torch.empty(f.shape[0], device=dev, dtype=torch.int32)
        check(L.ggs_visibility(f.shape[0], f.shape[0], v.shape[0], ptr(vv), ptr(ff), ptr(cam), ptr(tg), ptr(bd), ptr(scratch),
                               cap, ptr(mask), ptr(first), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ggs_visibility")
        outs.append((mask.clone(), first.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_full_size_speed_and_sanity():
    import time
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f = S.skirt_mesh()
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], 0), 0, device="cuda")
    cam = S.rig_cameras()[40].camera_center.cuda()
    mask = m.get_visible_mask(cam)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        mask = m.get_visible_mask(cam)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"\\nvisibility of 100k Gaussians vs 100k triangles: {dt * 1e3:.3f} ms, visible fraction {float(mask.float().mean()):.3f}")
    assert 0.3 < float(mask.float().mean()) < 0.7 and dt < 0.05
