"""Multi-GPU path on CPU: world_size-2 gloo process group, view sharding + one flat-bucket
all-reduce of the gradients (ggsplat.dist), exactly the code bench.py runs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ggsplat.dist import (all_reduce_bucket, all_reduce_densification_stats, all_reduce_grads, all_reduce_parts, bucket_views,
                          flatten_grads, shard_views, unflatten_into)


def test_shard_views_partition():
    for world in (1, 2, 4, 8):
        got = sorted(v for r in range(world) for v in shard_views(160, r, world))
        assert got == list(range(160))
        sizes = [len(shard_views(160, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert shard_views(5, 3, 8) == [3] and shard_views(5, 7, 8) == []


def test_flatten_roundtrip():
    ts = [torch.randn(5, 3), torch.randn(7), torch.randn(2, 4, 3)]
    flat = flatten_grads(ts)
    assert flat.numel() == 15 + 7 + 24
    out = [torch.zeros_like(t) for t in ts]
    unflatten_into(flat, out)
    assert all(torch.equal(a, b) for a, b in zip(ts, out))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # per-view "gradients": deterministic function of the view index, so the exact sum is known
    views = shard_views(10, rank, world)
    shapes = [(6, 3), (6, 1), (6, 1, 3), (4, 3)]
    grads = [torch.zeros(s) for s in shapes]
    for v in views:
        g = torch.Generator().manual_seed(100 + v)
        for t in grads:
            t += torch.randn(t.shape, generator=g)
    # the same exchange with the gradients living in the bucket from the start (what bench.py's captured step does)
    bucket = flatten_grads(grads)
    views = bucket_views(bucket, grads)
    all_reduce_bucket(bucket)
    flat = all_reduce_grads(grads, n_views_total=10, average=True)
    assert all(torch.allclose(v / 10.0, g, rtol=1e-6, atol=1e-7) for v, g in zip(views, grads))
    assert all(v.data_ptr() >= bucket.data_ptr() and v.shape == g.shape for v, g in zip(views, grads))
    acc, den, rad = torch.full((6, 1), float(rank + 1)), torch.ones(6, 1), torch.tensor([1.0 + rank, 5.0 - rank])
    all_reduce_densification_stats(acc, den, rad)
    # overlapped exchange (bench.py --overlap): the rank's views in two slices, slice 0 summed over the ranks while slice 1 is
    # "rendered"; against ONE all-reduce of the rank's whole bucket
    mine = shard_views(10, rank, world)

    def slice_bucket(vs):
        out = torch.zeros(sum(torch.Size(s).numel() for s in shapes))
        for v in vs:
            g = torch.Generator().manual_seed(100 + v)
            out += torch.cat([torch.randn(s, generator=g).reshape(-1) for s in shapes])
        return out
    half = (len(mine) + 1) // 2
    over = all_reduce_parts([lambda: slice_bucket(mine[:half]), lambda: slice_bucket(mine[half:])])
    single = slice_bucket(mine)
    all_reduce_bucket(single)
    # numpy arrays travel through the queue BY VALUE: torch tensors would be handed over as shared-memory file descriptors that
    # the parent can only open while this process is alive -- a race with the exit below (EOFError / ConnectionResetError /
    # FileNotFoundError in q.get())
    q.put((rank, [t.numpy().copy() for t in grads], flat.numel(), acc.numpy().copy(), den.numpy().copy(), rad.numpy().copy(),
           over.numpy().copy(), single.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


_last_failure = []


def _run_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
        as_t = torch.from_numpy
        res = [(r, [as_t(g) for g in grads], n, as_t(acc), as_t(den), as_t(rad), as_t(over), as_t(single))
               for r, grads, n, acc, den, rad, over, single in res]
        for p in procs:
            p.join(timeout=60)
        if any(p.exitcode != 0 for p in procs):
            _last_failure.append(f"exit codes {[p.exitcode for p in procs]}")
            return None
        return res
    except Exception as e:
        _last_failure.append(f"{type(e).__name__}: {e}; exit codes {[p.exitcode for p in procs]}")
        return None
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()


def test_two_rank_gradient_all_reduce_matches_single_process():
    res = None
    for _ in range(4):                      # a rendezvous can lose a port race on a busy box: retry with a new port
        res = _run_two_ranks()
        if res is not None:
            break
    assert res is not None, f"gloo world-size-2 run failed 4 times: {_last_failure}"
    shapes = [(6, 3), (6, 1), (6, 1, 3), (4, 3)]
    expect = [torch.zeros(s) for s in shapes]
    for v in range(10):
        g = torch.Generator().manual_seed(100 + v)
        for t in expect:
            t += torch.randn(t.shape, generator=g)
    expect = [t / 10.0 for t in expect]
    for rank, grads, n, acc, den, rad, over, single in res:
        assert n == sum(torch.Size(s).numel() for s in shapes)
        # the overlapped exchange == one all-reduce of the whole bucket, up to the order of the additions
        assert torch.allclose(over, single, rtol=1e-6, atol=1e-6) and over.shape == single.shape
        assert torch.allclose(over, torch.cat([t.reshape(-1) for t in expect]) * 10.0, rtol=1e-5, atol=1e-5)
        for a, b in zip(grads, expect):
            assert torch.allclose(a, b, atol=1e-6)
        assert torch.equal(acc, torch.full((6, 1), 3.0)) and torch.equal(den, torch.full((6, 1), 2.0))
        assert torch.equal(rad, torch.tensor([2.0, 5.0]))
    # both ranks end with identical gradients -> identical Adam steps without a parameter broadcast
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][6], res[1][6])        # ... through the overlapped exchange too


def test_all_reduce_parts_without_a_process_group_is_the_plain_sum():
    a, b = torch.arange(6.0), torch.ones(6)
    out = all_reduce_parts([lambda: a, lambda: b])
    assert torch.equal(out, a + b) and torch.equal(a, torch.arange(6.0))        # inputs untouched
    assert all_reduce_parts([lambda: a]) is a
