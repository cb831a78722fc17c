"""View sharding + gradient exchange for the multi-GPU inner step.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Parameters are replicated; rank r renders views r, r+G, r+2G, ...; after the local
fwd+bwd over its shard every rank holds a partial gradient, and ONE all-reduce(sum) of a single flat
fp32 bucket [mesh.v | _xyz | _features_dc | _features_rest | _opacity | _scaling | _rotation]
(5.6 MB at 100k Gaussians / SH degree 0) makes them identical again.  The collective is
latency-critical, not bandwidth-critical (<< 1 ms vs ~20+ ms of render work per step), so: one
bucket, one call per step, no overlap machinery.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """views[rank::world] -- round-robin so every rank gets a spread of the camera rings."""
    return list(range(rank, n_views, world))


def flatten_grads(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])


def unflatten_into(flat: torch.Tensor, tensors: Sequence[torch.Tensor]) -> None:
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n


def _distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_reduce_grads(grads: Sequence[torch.Tensor], n_views_total: int = 0, average: bool = False) -> torch.Tensor:
    """Sum (optionally / n_views_total) the gradient tensors over all ranks, in place, through one flat bucket.
    Returns the bucket (useful for tests).  World size 1 without averaging: nothing to do, no copies."""
    if not _distributed() and not (average and n_views_total > 0):
        return flatten_grads(grads)
    flat = flatten_grads(grads)
    if _distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average and n_views_total > 0:
        flat /= float(n_views_total)
    unflatten_into(flat, grads)
    return flat


def bucket_views(flat: torch.Tensor, like: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Views into a flat bucket shaped like `like` (no copies): gradients that live in the bucket from the start need
    neither the flatten nor the unflatten kernels around the collective."""
    out, o = [], 0
    for t in like:
        n = t.numel()
        out.append(flat[o:o + n].view(t.shape))
        o += n
    return out


def all_reduce_bucket(flat: torch.Tensor) -> None:
    """One all-reduce(sum) of a gradient bucket that is already flat (RCCL over xGMI; gloo in the CPU tests)."""
    if _distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def all_reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor):
    """Per-view statistics of the first-frame densification (scene/gaussian_model.py:410-412,
    s2_registration.py:314): sums for the gradient-norm accumulators, max for the screen radii."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
        dist.all_reduce(denom, op=dist.ReduceOp.SUM)
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)
