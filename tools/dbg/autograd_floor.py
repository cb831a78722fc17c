"""What a backward() through PyTorch's autograd engine costs on this host before any of this repo's code runs: wall time per
forward + backward of (a) one built-in op, (b) one custom autograd.Function whose forward / backward launch nothing, (c) the same
with seven leaf inputs (the parameter tensors of a mesh-bound model).  Host-side rates: the GPU work is one tiny kernel."""
import time, torch
dev = "cuda"
x = torch.randn(100000, 3, device=dev, requires_grad=True)
leaves = [torch.randn(100000, 3, device=dev, requires_grad=True) for _ in range(7)]
g = torch.ones(100000, 3, device=dev)


class Nop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        return ts[0].detach()

    @staticmethod
    def backward(ctx, go):
        return (go,) + (None,) * 6 if len(ctx.needs_input_grad) == 7 else (go,)


def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def builtin():
    x.grad = None
    (x * 2.0).backward(g)


def custom1():
    x.grad = None
    Nop.apply(x).backward(g)


def custom7():
    for t in leaves: t.grad = None
    Nop.apply(*leaves).backward(g)


print(f"one built-in op (mul) forward + backward(grad): {timed(builtin):7.1f} us")
print(f"one custom Function, no launches, one leaf:      {timed(custom1):7.1f} us")
print(f"one custom Function, no launches, seven leaves:  {timed(custom7):7.1f} us")
