"""Times the two passes of the fused photometric loss (plain and region-of-interest forms) at 1080p with HIP events:
one view (the launch shape of the s2 iteration) and 16 views.  GGS_LIB_PATH selects the library build."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat._lib import LIB_PATH, ptr
L = C.CDLL(LIB_PATH)            # bound by hand: libraries of earlier builds (without the newest entry points) can be timed too
L.ggs_photometric_scratch_bytes.restype = C.c_size_t
P_ = C.c_void_p
for n_, k_ in (("ggs_photometric_forward", 6), ("ggs_photometric_backward", 7), ("ggs_photometric_forward_roi", 9), ("ggs_photometric_backward_roi", 10),
               ("ggs_photometric_forward_sparse", 11), ("ggs_mask_tiles", 3)):
    if hasattr(L, n_):
        getattr(L, n_).argtypes = [C.c_int] * 3 + [P_] * k_


def check(rc, what):
    assert rc == 0, what


H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
for V in (1, 16):
    img, gt = torch.rand(V, 3, H, W, generator=g).cuda(), torch.rand(V, 3, H, W, generator=g).cuda()
    mask = (torch.rand(V, 1, H, W, generator=g) > 0.2).float().cuda()
    scratch = torch.empty(L.ggs_photometric_scratch_bytes(V, H, W), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(V, 2, device="cuda"); w = torch.tensor([[0.8, -0.2]] * V, device="cuda"); d = torch.empty_like(img)
    tgx, tgy = (W + 15) // 16, (H + 15) // 16
    tc = torch.zeros(V, tgy, tgx, dtype=torch.int32, device="cuda"); tc[:, 20:50, 40:80] = 5      # ~14 % of the tiles have a list
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # silhouette mask: the tiles with a list, grown by a margin (a segmentation of the garment), ~16 % of the frame
    sil = torch.zeros(V, 1, H, W, device="cuda"); sil[:, :, 20 * 16 - 24:50 * 16 + 24, 40 * 16 - 24:80 * 16 + 24] = 1.0
    mt = torch.empty(V, tgy * tgx, dtype=torch.int32, device="cuda")
    jobs = {
        "forward": lambda: check(L.ggs_photometric_forward(V, H, W, ptr(img), ptr(gt), ptr(mask), ptr(sums), ptr(scratch), s), "f"),
        "backward": lambda: check(L.ggs_photometric_backward(V, H, W, ptr(img), ptr(gt), ptr(mask), ptr(scratch), ptr(w), ptr(d), s), "b"),
        "forward_roi": lambda: check(L.ggs_photometric_forward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(sums), ptr(scratch), s), "fr"),
        "backward_roi": lambda: check(L.ggs_photometric_backward_roi(V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(scratch), ptr(w), ptr(d), s), "br"),
        "forward_roi(silhouette)": lambda: check(L.ggs_photometric_forward_roi(V, H, W, ptr(img), ptr(gt), ptr(sil), None, None, ptr(tc), ptr(sums), ptr(scratch), s), "frs"),
    }
    if hasattr(L, "ggs_photometric_forward_sparse"):
        check(L.ggs_mask_tiles(V, H, W, ptr(sil), ptr(mt), s), "mt")
        jobs["forward_sparse(silhouette)"] = lambda: check(L.ggs_photometric_forward_sparse(
            V, H, W, ptr(img), ptr(gt), ptr(sil), None, None, ptr(tc), ptr(mt), None, ptr(sums), ptr(scratch), s), "fs")
        jobs["backward_roi(silhouette)"] = lambda: check(L.ggs_photometric_backward_roi(
            V, H, W, ptr(img), ptr(gt), ptr(sil), None, None, ptr(tc), ptr(scratch), ptr(w), ptr(d), s), "brs")
        jobs["mask_tiles"] = lambda: check(L.ggs_mask_tiles(V, H, W, ptr(sil), ptr(mt), s), "mt")
        mt_d = torch.empty_like(mt); check(L.ggs_mask_tiles(V, H, W, ptr(mask), ptr(mt_d), s), "mtd")
        jobs["forward_sparse(dense mask)"] = lambda: check(L.ggs_photometric_forward_sparse(
            V, H, W, ptr(img), ptr(gt), ptr(mask), None, None, ptr(tc), ptr(mt_d), None, ptr(sums), ptr(scratch), s), "fsd")
    out = []
    for name, fn in jobs.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        n = 40 if V == 1 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) / n / V * 1e3:7.1f}")
    print(f"V={V:2d}  us per view:  " + "   ".join(out))
