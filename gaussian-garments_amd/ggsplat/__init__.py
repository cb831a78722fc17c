"""ggsplat -- host side of the MI355X-native Gaussian-splat rasterizer.

Sub-modules are imported lazily by their users; importing this package does not
load the HIP library (``ggsplat._lib`` does, and raises if it is missing).
"""
__all__ = ["cameras", "synthetic"]
