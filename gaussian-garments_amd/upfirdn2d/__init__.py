"""Drop-in for the reference's extension module `upfirdn2d` (scene/styleunet/upfirdn2d.py:30
`import upfirdn2d as upfirdn2d_op`; built from scene/styleunet/upfirdn2d*.{cpp,cu}).  One symbol: upfirdn2d.
float / half / double are processed natively (no host-side conversion); the FIR kernel is brought to the input's dtype
like the upstream op requires of its caller."""
import ctypes as C

import torch

from ggsplat._lib import check, dtype_code, lib, ptr


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """input [major, in_h, in_w, minor], kernel [kh, kw] -> [major, out_h, out_w, minor]."""
    if input.device.type != "cuda":
        raise RuntimeError("input must be a CUDA tensor")
    x = input.contiguous()
    code = dtype_code(x.dtype)
    k = kernel.to(device=x.device, dtype=x.dtype).contiguous()
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    oh, ow = C.c_int(), C.c_int()
    L = lib()
    check(L.ggs_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                                   C.byref(oh), C.byref(ow)), "ggs_upfirdn2d_out_size")
    out = torch.empty(major, max(oh.value, 0), max(ow.value, 0), minor, device=x.device, dtype=x.dtype)
    check(L.ggs_upfirdn2d_t(code, major, in_h, in_w, minor, ptr(x), ptr(k), kh, kw, up_x, up_y, down_x, down_y, pad_x0,
                            pad_x1, pad_y0, pad_y1, ptr(out), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)),
          "ggs_upfirdn2d")
    return out
