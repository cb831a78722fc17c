"""View sharding + gradient exchange for the multi-GPU inner step.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Parameters are replicated; rank r renders views r, r+G, r+2G, ...; after the local
fwd+bwd over its shard every rank holds a partial gradient, and ONE all-reduce(sum) of a single flat
fp32 bucket [mesh.v | _xyz | _features_dc | _features_rest | _opacity | _scaling | _rotation]
(5.6 MB at 100k Gaussians / SH degree 0) makes them identical again.  The collective is
latency-critical, not bandwidth-critical: one bucket, one call (`all_reduce_bucket`).  At 8 GPUs a rank's
step is ~1.5 ms and a 6.2 MB ring all-reduce over xGMI is 0.1-0.3 ms of it, fully exposed behind the compute:
`all_reduce_parts` splits the rank's views into slices and sums slice i over the ranks WHILE slice i + 1 is
being rendered (the sum over views is linear), so only the last slice's collective is exposed.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

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


def all_reduce_parts(parts: Sequence[Callable[[], torch.Tensor]]) -> torch.Tensor:
    """Overlapped gradient exchange.  `parts[i]()` renders the i-th slice of this rank's views and returns its flat gradient
    bucket (every call the same layout); the all-reduce of slice i is started asynchronously (RCCL on its own stream, ordered
    behind the producer by an event) and runs while slice i + 1 is computed; the slices are added up at the end.  Returns the
    bucket summed over ALL views of ALL ranks -- equal to one all-reduce of the sum up to the order of the additions.
    One part, or no process group: the plain single all-reduce."""
    flats, works = [], []
    for fn in parts:
        flat = fn()
        flats.append(flat)
        if _distributed():
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    total = flats[0]
    for f in flats[1:]:
        total = total + f            # not in place: the parts may be the static outputs of captured graphs
    return total


def all_reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor):
    """Per-view statistics of the first-frame densification (scene/gaussian_model.py:410-412,
    s2_registration.py:314): sums for the gradient-norm accumulators, max for the screen radii."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM)
        dist.all_reduce(denom, op=dist.ReduceOp.SUM)
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX)
