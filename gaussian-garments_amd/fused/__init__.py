"""Drop-in for the reference's extension module `fused` (scene/styleunet/fused_act.py:30 `import fused`;
built by setup.sh:29-30 from scene/styleunet/fused_bias_act*.{cpp,cu}).  One symbol: fused_bias_act.
float / half / double are processed natively (no host-side conversion)."""
import ctypes as C

import torch

from ggsplat._lib import check, dtype_code, lib, ptr


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """y = act(input + bias[channel]) * scale; empty `bias` / `refer` tensors mean "absent" (upstream convention)."""
    if input.device.type != "cuda":
        raise RuntimeError("input must be a CUDA tensor")
    x = input.contiguous()
    code = dtype_code(x.dtype)
    b = bias.to(dtype=x.dtype).contiguous() if bias is not None and bias.numel() else None
    r = refer.to(dtype=x.dtype).contiguous() if refer is not None and refer.numel() else None
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    y = torch.empty_like(x)
    check(lib().ggs_fused_bias_act_t(code, x.numel(), ptr(x), ptr(b), ptr(r), step_b, b.numel() if b is not None else 1,
                                     int(act), int(grad), float(alpha), float(scale), ptr(y),
                                     C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "ggs_fused_bias_act")
    return y
