#!/usr/bin/env python
"""Timings of the "next" rows that have no line in bench.py, against the roofline that bounds each (run on the
GPU box; prints a markdown table -> profiles/rNN_next_rows.md):
  f1  fused L1 + SSIM loss, value + dL/dimage (HBM: two tile passes, 136 B per pixel incl. the derivative maps)
  f2  distCUDA2 (exact brute-force 3-NN: fp32 VALU, 8 flop per pair)
  f4  GPU visibility (first-hit ray cast against the mesh), guarded Adam (HBM: 28 B per parameter)
  a6-a9 fused mesh binding forward / backward (HBM: ~100 B / ~150 B per Gaussian)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
import torch  # noqa: E402
from ggsplat import _lib, synthetic as S  # noqa: E402
from ggsplat.adam import GraphAdam  # noqa: E402
from ggsplat.loss import fused_photometric_loss  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel, mesh_bind  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402


def timed(fn, reps=20):
    # three untimed calls: the first call after a host-only stretch finds the GPU at idle clocks (brute-force 3-NN: 7.0 ms for
    # the first call, 4.2 for calls 2-5, 4.06 steady -- tools/dbg/time_knn.py; r04_next_rows.md quoted 13.4 ms from one warm-up call
    # and five timed ones behind the mesh construction on the host: a measurement artefact, the kernel had not changed)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = "cuda"
    print(f"# 'Next' rows of SURVEY 8f on one MI355X (library build {_lib.build_id()})\n")
    print("| row | op | size | ms | achieved | roofline | % |\n|---|---|---|---|---|---|---|")
    H, W = 1080, 1920
    for V in (1, 32):
        img = torch.rand(V, 3, H, W, device=dev, requires_grad=True)
        gt, mask = torch.rand(V, 3, H, W, device=dev), (torch.rand(V, 1, H, W, device=dev) > 0.1).float()

        def loss():
            img.grad = None
            a, b = fused_photometric_loss(img, gt, mask, 0.2)
            (a.sum() + b.sum()).backward()
        t = timed(loss)
        gb = 136.0 * V * H * W / t / 1e9
        print(f"| f1 | fused L1+SSIM value + gradient, autograd op | {V} x 3x1080x1920 | {t*1e3:.3f} | {gb:.0f} GB/s | 8000 GB/s HBM | {gb/80:.1f} |")
        # the two C entry points alone (what the graph-replayed steps issue): no autograd / allocator time on the host
        import ctypes as C
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        scratch = torch.empty(L.ggs_photometric_scratch_bytes(V, H, W), device=dev, dtype=torch.uint8)
        sums, wts, dimg = torch.zeros(V, 2, device=dev), torch.tensor([[0.8, -0.2]] * V, device=dev), torch.empty(V, 3, H, W, device=dev)
        x = img.detach()

        def loss_c():
            _lib.check(L.ggs_photometric_forward(V, H, W, _lib.ptr(x), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(sums), _lib.ptr(scratch), stream), "fwd")
            _lib.check(L.ggs_photometric_backward(V, H, W, _lib.ptr(x), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(scratch), _lib.ptr(wts), _lib.ptr(dimg), stream), "bwd")
        t = timed(loss_c)
        gb = 136.0 * V * H * W / t / 1e9
        print(f"| f1 | fused L1+SSIM value + gradient, the two C entry points | {V} x 3x1080x1920 | {t*1e3:.3f} | {gb:.0f} GB/s | 8000 GB/s HBM | {gb/80:.1f} |")
    v, f = S.skirt_mesh()
    P = f.shape[0]
    pts = v[f].mean(1).to(dev)
    t = timed(lambda: distCUDA2(pts, brute_force=True), reps=20)
    tf = 8.0 * P * P / t / 1e12
    print(f"| f2 | distCUDA2, exact 3-NN, brute force ({P} face centres of the skirt) | {P}^2 pairs | {t*1e3:.3f} | {tf:.1f} TFLOP/s fp32 | 157.3 TFLOP/s fp32 vector | {tf/1.573:.1f} |")
    t = timed(lambda: distCUDA2(pts), reps=10)
    print(f"| f2 | distCUDA2, exact 3-NN, uniform grid (the default; same bits) | {P} points | {t*1e3:.3f} | {P/t/1e6:.0f} M points/s | -- | -- |")
    v5, f5 = S.skirt_mesh(500, 500)
    pts5 = v5[f5].mean(1).to(dev)
    t = timed(lambda: distCUDA2(pts5), reps=5)
    print(f"| f2 | distCUDA2, uniform grid, config 5 ({pts5.shape[0]} face centres) | {pts5.shape[0]} points | {t*1e3:.3f} | {pts5.shape[0]/t/1e6:.0f} M points/s | -- | -- |")
    ptr_ = torch.randn(500000, 3, device=dev)
    t = timed(lambda: distCUDA2(ptr_), reps=5)
    print(f"| f2 | distCUDA2, uniform grid, 500000 normally distributed points | 500000 points | {t*1e3:.3f} | {0.5/t:.0f} M points/s | -- | -- |")
    params = S.skirt_gaussian_params(P, sh_degree=0)
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device=dev)
    cam = S.rig_cameras()[5].camera_center.to(dev)
    t = timed(lambda: m.get_visible_mask(cam), reps=10)
    print(f"| f4 | GPU visibility: {P} rays against {P} triangles (grid + exact first hit) | -- | {t*1e3:.3f} | {P/t/1e6:.0f} M rays/s | -- | -- |")
    ps = [torch.randn(n, device=dev, requires_grad=True) for n in (150600, 300000, 300000, 100000, 300000, 400000)]
    opt = GraphAdam([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15)
    for p in ps:
        p.grad = torch.randn_like(p)
    t = timed(lambda: opt.step())
    n = sum(p.numel() for p in ps)
    gb = 28.0 * n / t / 1e9
    print(f"| f4 | guarded Adam, one launch for the 7 parameter groups of config 2 | {n} parameters | {t*1e3:.3f} | {gb:.0f} GB/s | 8000 GB/s HBM | {gb/80:.1f} |")
    lx = params["_xyz"].to(dev).requires_grad_(True); ls = params["_scaling"].to(dev).requires_grad_(True)
    lr_ = params["_rotation"].to(dev).requires_grad_(True); vv = v.to(dev).requires_grad_(True)
    ff, bb = f.to(dev), params["binding"].to(dev)
    t = timed(lambda: mesh_bind(vv.detach(), ff, bb, lx.detach(), ls.detach(), lr_.detach()))
    gb = 100.0 * P / t / 1e9
    print(f"| a6-a9 | fused mesh binding forward | {P} Gaussians | {t*1e3:.3f} | {gb:.0f} GB/s | 8000 GB/s HBM (latency-bound at this size) | {gb/80:.1f} |")
    g = [torch.randn(P, 3, device=dev), torch.randn(P, 3, device=dev), torch.randn(P, 4, device=dev)]

    def bwd():
        for x in (vv, lx, ls, lr_):
            x.grad = None
        torch.autograd.backward(list(mesh_bind(vv, ff, bb, lx, ls, lr_)), g)
    t = timed(bwd)
    print(f"| a6-a9 | fused mesh binding forward + backward (autograd op) | {P} Gaussians | {t*1e3:.3f} | -- | -- | -- |")


if __name__ == "__main__":
    main()
