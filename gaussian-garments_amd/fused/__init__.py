"""Drop-in for the reference's extension module `fused` (scene/styleunet/fused_act.py:30 `import fused`;
built by setup.sh:29-30 from scene/styleunet/fused_bias_act*.{cpp,cu}).  One symbol: fused_bias_act."""
import ctypes as C

import torch

from ggsplat._lib import check, lib, ptr


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """y = act(input + bias[channel]) * scale; empty `bias` / `refer` tensors mean "absent" (upstream convention)."""
    if input.device.type != "cuda":
        raise RuntimeError("input must be a CUDA tensor")
    x = input.contiguous()
    dt = x.dtype
    xf = x.float()
    b = bias.contiguous().float() if bias is not None and bias.numel() else None
    r = refer.contiguous().float() if refer is not None and refer.numel() else None
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    y = torch.empty_like(xf)
    check(lib().ggs_fused_bias_act(xf.numel(), ptr(xf), ptr(b), ptr(r), step_b, b.numel() if b is not None else 1,
                                   int(act), int(grad), float(alpha), float(scale), ptr(y),
                                   C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "ggs_fused_bias_act")
    return y if dt == torch.float32 else y.to(dt)
