"""`from simple_knn._C import distCUDA2` -- HIP implementation in libggsplat.so (csrc/ggs_knn.hip)."""
import ctypes as C

import torch

from ggsplat._lib import check, lib, ptr


def distCUDA2(points: torch.Tensor, brute_force: bool = False) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> [P] mean squared distance to the 3 nearest neighbours (self excluded).
    Grid search (ggs_dist2_3nn_grid) by default; brute_force=True runs the O(P^2) kernel (same result bit for bit)."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 expects a GPU tensor (as the reference passes it)")
    pts = points.detach().float().contiguous()
    P = pts.shape[0]
    out = torch.empty(P, device=pts.device, dtype=torch.float32)
    stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
    L = lib()
    if brute_force or P < 64:
        check(L.ggs_dist2_3nn(P, ptr(pts), ptr(out), stream), "ggs_dist2_3nn")
    else:
        scratch = torch.empty(L.ggs_dist2_3nn_scratch_bytes(P), device=pts.device, dtype=torch.uint8)
        check(L.ggs_dist2_3nn_grid(P, ptr(pts), ptr(out), ptr(scratch), stream), "ggs_dist2_3nn_grid")
    return out
