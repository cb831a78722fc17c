#!/usr/bin/env python
"""GB/s of the StyleGAN2 ops at StyleUNet shapes against the 8 TB/s HBM roofline (run on the GPU box):
algorithmic bytes = read the input once + write the output once (+ the reference tensor for the derivative mode).
Prints a markdown table (-> profiles/rNN_stylegan_ops.md)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import fused  # noqa: E402
import upfirdn2d as U  # noqa: E402
from ggsplat import _lib  # noqa: E402


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    k = torch.tensor([1.0, 3.0, 3.0, 1.0]); k = (k[None] * k[:, None]); k = (k / k.sum()).cuda()
    h = torch.tensor([[0.5, 0.5], [0.5, 0.5]]).cuda()
    print(f"# StyleGAN2 ops at StyleUNet shapes (MI355X, library build {_lib.build_id()})\n")
    print("Algorithmic bytes = input once + output once; peak 8000 GB/s.  `generic` = the same configuration forced through "
          "the one-lane-per-output gather kernel (minor = 2 trick: two planes interleaved), for comparison.\n")
    print("| op | dtype | shape [major,H,W] | ms | GB/s | % of 8 TB/s |\n|---|---|---|---|---|---|")
    for dt in (torch.float32, torch.float16):
        es = torch.empty(0, dtype=dt).element_size()
        for (major, H, W) in ((64, 1024, 1024), (16, 2048, 2048)):
            x = torch.randn(major, H, W, 1, device="cuda").to(dt)
            cases = [("blur 4x4 pad(2,1)", k, 1, 1, (2, 1, 2, 1)), ("upsample 4x4 up2", k * 4, 2, 1, (2, 1, 2, 1)),
                     ("downsample 4x4 down2", k, 1, 2, (1, 1, 1, 1)), ("haar 2x2 down2", h, 1, 2, (0, 0, 0, 0)),
                     ("inverse haar 2x2 up2", h, 2, 1, (1, 0, 1, 0))]
            for name, kk, up, down, pad in cases:
                kk = kk.to(dt)
                out = U.upfirdn2d(x, kk, up, up, down, down, *pad)
                t = timed(lambda: U.upfirdn2d(x, kk, up, up, down, down, *pad))
                gb = (x.numel() + out.numel()) * es / t / 1e9
                print(f"| upfirdn2d {name} | {str(dt)[6:]} | {major}x{H}x{W} | {t*1e3:.3f} | {gb:.0f} | {gb/80:.1f} |")
            xc = torch.randn(major // 4, 4 * 16, H // 4, W // 4, device="cuda").to(dt) if False else torch.randn(4, major * 4, H // 4, W // 4, device="cuda").to(dt)
            b = torch.randn(xc.shape[1], device="cuda").to(dt)
            e = xc.new_empty(0)
            y = fused.fused_bias_act(xc, b, e, 3, 0, 0.2, 2 ** 0.5)
            t = timed(lambda: fused.fused_bias_act(xc, b, e, 3, 0, 0.2, 2 ** 0.5))
            gb = 2 * xc.numel() * es / t / 1e9
            print(f"| fused_bias_act fwd (bias + lrelu) | {str(dt)[6:]} | {tuple(xc.shape)} | {t*1e3:.3f} | {gb:.0f} | {gb/80:.1f} |")
            t = timed(lambda: fused.fused_bias_act(xc, e, y, 3, 1, 0.2, 2 ** 0.5))
            gb = 3 * xc.numel() * es / t / 1e9
            print(f"| fused_bias_act grad (ref) | {str(dt)[6:]} | {tuple(xc.shape)} | {t*1e3:.3f} | {gb:.0f} | {gb/80:.1f} |")
    # generic kernel on the same blur for comparison (minor = 2 forces it)
    x2 = torch.randn(32, 1024, 1024, 2, device="cuda")
    o2 = U.upfirdn2d(x2, k, 1, 1, 1, 1, 2, 1, 2, 1)
    t = timed(lambda: U.upfirdn2d(x2, k, 1, 1, 1, 1, 2, 1, 2, 1))
    gb = (x2.numel() + o2.numel()) * 4 / t / 1e9
    print(f"| upfirdn2d blur 4x4, generic kernel (minor = 2) | float32 | 32x1024x1024x2 | {t*1e3:.3f} | {gb:.0f} | {gb/80:.1f} |")


if __name__ == "__main__":
    main()
