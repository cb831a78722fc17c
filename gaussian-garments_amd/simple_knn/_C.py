"""`from simple_knn._C import distCUDA2` -- HIP implementation in libggsplat.so (csrc/ggs_knn.hip)."""
import ctypes as C

import torch

from ggsplat._lib import check, lib, ptr


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> [P] mean squared distance to the 3 nearest neighbours."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 expects a GPU tensor (as the reference passes it)")
    pts = points.detach().float().contiguous()
    out = torch.empty(pts.shape[0], device=pts.device, dtype=torch.float32)
    check(lib().ggs_dist2_3nn(pts.shape[0], ptr(pts), ptr(out),
                              C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)), "ggs_dist2_3nn")
    return out
