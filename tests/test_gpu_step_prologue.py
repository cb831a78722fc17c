"""ggs_step_prologue (zero fills | block copy | sigmoid | mesh binding in one launch), its pre-clear marks, the step's result
block (ggs_registration_aux_tail) and the Adam update that advances its own step count (ggs_adam_tick_step_multi)."""
import ctypes as C

import pytest
import torch

from ggsplat import synthetic as S

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _mesh_case(seed=3):
    v, f = S.skirt_mesh(24, 17, seed=seed)
    g = torch.Generator().manual_seed(seed)
    binding = torch.randint(0, f.shape[0], (f.shape[0] * 2 + 37,), generator=g)
    P = binding.shape[0]
    local, ls, rr = torch.randn(P, 3, generator=g) * 0.3, torch.randn(P, 3, generator=g) * 0.3 - 0.5, torch.randn(P, 4, generator=g)
    b = torch.rand(P, 3, generator=g)
    return [t.cuda() for t in (v, f, binding, local, ls, rr, b / b.sum(1, keepdim=True))]


def test_prologue_jobs_match_the_separate_launches():
    from ggsplat._lib import GgsStepPrologue, check, host_mapped_pointer, lib, ptr
    L = lib()
    v, f, binding, local, ls, rr, bary = _mesh_case()
    P, Fn = binding.shape[0], f.shape[0]
    # reference: the separate entry points / torch
    xyz0, sc0, rot0 = torch.empty_like(local), torch.empty_like(ls), torch.empty_like(rr)
    check(L.ggs_mesh_bind_forward(P, Fn, ptr(v), ptr(f), ptr(binding), ptr(local), ptr(ls), ptr(rr), ptr(bary), ptr(xyz0),
                                  ptr(sc0), ptr(rot0), _stream()), "ggs_mesh_bind_forward")
    logit = torch.randn(P, 1, device="cuda") * 4
    # ranges with every alignment / tail: garbage in, guard words around them must survive
    buf = torch.full((4096 + 64,), 7.0, device="cuda")
    ranges = [(1, 5), (16, 1024), (1100, 3), (1203, 777), (2000, 4)]        # (first word, words)
    big = torch.full((300_003,), 3.0, device="cuda")
    host = torch.arange(44, dtype=torch.float32).pin_memory()
    dst = torch.zeros(44, device="cuda")
    pro = GgsStepPrologue()
    pro.n_clear = len(ranges) + 1
    for i, (a, n) in enumerate(ranges):
        pro.clear_ptr[i], pro.clear_bytes[i] = buf.data_ptr() + 4 * a, 4 * n
    pro.clear_ptr[len(ranges)], pro.clear_bytes[len(ranges)] = big.data_ptr() + 4, 4 * (big.numel() - 2)
    mapped = host_mapped_pointer(host)
    assert mapped, "PyTorch's pinned memory is expected to be mapped into the device's address space on ROCm"
    pro.copy_src, pro.copy_dst, pro.copy_bytes = mapped, dst.data_ptr(), 176
    pro.P, pro.F = P, Fn
    pro.verts, pro.faces, pro.binding = ptr(v), ptr(f), ptr(binding)
    pro.local_xyz, pro.log_scaling, pro.raw_rot, pro.bary = ptr(local), ptr(ls), ptr(rr), ptr(bary)
    xyz, sc, rot, op = torch.empty_like(local), torch.empty_like(ls), torch.empty_like(rr), torch.empty_like(logit)
    pro.xyz, pro.scaling, pro.rotation = ptr(xyz), ptr(sc), ptr(rot)
    pro.n_opacity, pro.opacity_logit, pro.opacity = P, ptr(logit), ptr(op)
    check(L.ggs_step_prologue(C.byref(pro), _stream()), "ggs_step_prologue")
    torch.cuda.synchronize()
    assert torch.equal(xyz, xyz0) and torch.equal(sc, sc0) and torch.equal(rot, rot0)
    assert torch.allclose(op, torch.sigmoid(logit), rtol=2e-7, atol=0)
    expect = torch.full_like(buf, 7.0)
    for a, n in ranges:
        expect[a:a + n] = 0
    assert torch.equal(buf, expect)
    assert float(big[0]) == 3.0 and float(big[-1]) == 3.0 and not bool(big[1:-1].any())
    assert torch.equal(dst.cpu(), host)
    # argument errors, not memory faults
    bad = GgsStepPrologue()
    bad.n_clear = 1
    bad.clear_ptr[0], bad.clear_bytes[0] = buf.data_ptr() + 2, 8
    assert L.ggs_step_prologue(C.byref(bad), _stream()) != 0 and b"aligned" in L.ggs_last_error()
    bad = GgsStepPrologue()
    bad.n_clear = 9
    assert L.ggs_step_prologue(C.byref(bad), _stream()) != 0


def test_preclear_marks_are_consumed_once_and_dropped_by_the_next_prologue():
    from ggsplat._lib import GgsStepPrologue, check, lib, ptr
    L = lib()
    H, W = 48, 64
    g = torch.Generator().manual_seed(0)
    img, gt = torch.rand(1, 3, H, W, generator=g).cuda(), torch.rand(1, 3, H, W, generator=g).cuda()
    scratch = torch.empty(L.ggs_photometric_scratch_bytes(1, H, W), dtype=torch.uint8, device="cuda")
    sums = torch.empty(2, device="cuda")

    def loss_sums():
        check(L.ggs_photometric_forward(1, H, W, ptr(img), ptr(gt), None, ptr(sums), ptr(scratch), _stream()), "ggs_photometric_forward")
        return sums.clone()

    sums.fill_(123.0)
    ref = loss_sums()                                     # no mark: the call clears the sums itself
    assert float(ref[0]) > 0 and float(ref[0]) < 3 * H * W

    def prologue(marked, has_consumer=True):
        pro = GgsStepPrologue()
        if marked:
            pro.n_clear = 1
            pro.clear_ptr[0], pro.clear_bytes[0] = sums.data_ptr(), 8
            pro.consumer_mask = 1 if has_consumer else 0
        check(L.ggs_step_prologue(C.byref(pro), _stream()), "ggs_step_prologue")

    prologue(True)
    assert not bool(sums.any())                           # the prologue cleared them ...
    sums.fill_(10.0)                                      # (breaking the contract on purpose: shows that the call's own fill is skipped)
    assert torch.allclose(loss_sums(), ref + 10.0, rtol=1e-6)
    sums.fill_(10.0)
    assert torch.allclose(loss_sums(), ref, rtol=1e-6)    # ... once: the mark is gone, the call clears again
    prologue(True)
    prologue(False)                                       # a later prologue drops marks nobody consumed
    sums.fill_(10.0)
    assert torch.allclose(loss_sums(), ref, rtol=1e-6)
    # another stream does not consume this stream's mark
    prologue(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sums.fill_(10.0)
        got = loss_sums()
    torch.cuda.current_stream().wait_stream(side)
    assert torch.allclose(got, ref, rtol=1e-6)
    prologue(False)
    # a range the caller did not flag as having a consumer is cleared but NOT marked (dL/dvertices)
    prologue(True, has_consumer=False)
    assert not bool(sums.any())
    sums.fill_(10.0)
    assert torch.allclose(loss_sums(), ref, rtol=1e-6)
    # marks end with the step: ggs_step_end() ...
    prologue(True)
    check(L.ggs_step_end(), "ggs_step_end")
    sums.fill_(10.0)
    assert torch.allclose(loss_sums(), ref, rtol=1e-6)
    # ... and ggs_registration_aux*, the step's last consumer, drops what is left (here: a mark it does not consume itself)
    prologue(True)
    P = 64
    radii = torch.ones(P, dtype=torch.int32, device="cuda")
    check(L.ggs_registration_aux(P, None, None, ptr(radii), None, None, None, None, 0.0, 0.0, 0.0, 0.0, None, None, None, None,
                                 None, None, None, None, _stream()), "ggs_registration_aux")
    sums.fill_(10.0)
    assert torch.allclose(loss_sums(), ref, rtol=1e-6)


def test_step_clear_plan_names_what_forward_and_backward_clear():
    from ggsplat import rasterizer as R
    ws = R.plan_step(5000, 1, 0, 320, 240, 2, torch.device("cuda", 0))
    assert ws.scratch_clear == 2 * 5000 * 48 and ws.scratch.numel() >= ws.scratch_clear
    T = ((320 + 15) // 16) * ((240 + 15) // 16)
    assert 16 + 2 * T * 4 * 2 <= ws.bin_clear <= ws.bin.numel()


def test_adam_update_that_ticks_itself_matches_tick_then_update():
    from ggsplat._lib import check, lib
    L = lib()
    sizes = [7, 4096, 300_001, 12]                         # one workgroup, several, capped grid, tail
    g = torch.Generator().manual_seed(1)
    nb = int(L.ggs_adam_state_bytes())

    def fresh():
        return ([torch.randn(n, generator=g).cuda() for n in sizes], [torch.zeros(n, device="cuda") for n in sizes],
                [torch.zeros(n, device="cuda") for n in sizes], [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in sizes])

    pa, ma, va, sa = fresh()
    pb, mb, vb, sb = [t.clone() for t in pa], [t.clone() for t in ma], [t.clone() for t in va], [t.clone() for t in sa]
    lr = torch.full((len(sizes),), 1e-2, device="cuda")
    guard = torch.zeros(1, dtype=torch.int64, device="cuda")
    n = len(sizes)
    numel = (C.c_size_t * n)(*sizes)

    def col(ts):
        return (C.c_void_p * n)(*[t.data_ptr() for t in ts])

    lrs = (C.c_void_p * n)(*[lr.data_ptr() + 4 * i for i in range(n)])
    for it in range(5):
        grads = [torch.randn(k, generator=g).cuda() for k in sizes]
        guard.fill_(1 if it == 2 else 0)                   # a guarded step changes nothing, not even the step count
        check(L.ggs_adam_tick_multi(n, col(sa), 0.9, 0.999, guard.data_ptr(), _stream()), "tick")
        check(L.ggs_adam_step_multi(n, numel, col(pa), col(grads), col(ma), col(va), lrs, col(sa), 0.9, 0.999, 1e-15,
                                    guard.data_ptr(), _stream()), "step")
        check(L.ggs_adam_tick_step_multi(n, numel, col(pb), col(grads), col(mb), col(vb), lrs, col(sb), 0.9, 0.999, 1e-15,
                                         guard.data_ptr(), _stream()), "tick_step")
    torch.cuda.synchronize()
    for a, b in zip(pa + ma + va, pb + mb + vb):
        assert torch.equal(a, b)
    for a, b in zip(sa, sb):
        assert torch.equal(a[:32], b[:32]) and int(a[:8].view(torch.int64)) == 4
        assert not bool(b[32:].any())                      # the workgroup ticket is back to zero


def test_registration_aux_tail_writes_the_result_block_and_ticks_the_optimiser():
    from ggsplat._lib import GgsStepTail, check, host_mapped_pointer, lib, ptr
    L = lib()
    P = 1000
    g = torch.Generator().manual_seed(2)
    xyz, ls = (torch.randn(P, 3, generator=g) * 0.5).cuda(), (torch.randn(P, 3, generator=g) * 0.3 - 1).cuda()
    radii = (torch.rand(P, generator=g) > 0.3).int().cuda()
    m2d, op, dop = torch.randn(P, 3, generator=g).cuda(), torch.rand(P, 1, generator=g).cuda(), torch.randn(P, 1, generator=g).cuda()
    sums = torch.tensor([12.5, 99.0], device="cuda")
    hdr = torch.tensor([4242, 0], dtype=torch.int64, device="cuda")
    nb = int(L.ggs_adam_state_bytes())

    def run(tail_states, out_ptr, guard_value):
        hdr[1] = guard_value
        dx, dl, dlogit = torch.zeros(P, 3, device="cuda"), torch.zeros(P, 3, device="cuda"), torch.empty(P, 1, device="cuda")
        losses, scratch = torch.zeros(3, device="cuda"), torch.zeros(4, device="cuda")
        tail = None
        if out_ptr:
            tail = GgsStepTail(ptr(sums), ptr(hdr), out_ptr)
            tail.n_adam_states = len(tail_states)
            for i, t in enumerate(tail_states):
                tail.adam_states[i] = t.data_ptr()
            tail.beta1, tail.beta2 = 0.9, 0.999
        check(L.ggs_registration_aux_tail(P, ptr(xyz), ptr(ls), ptr(radii), ptr(m2d), ptr(op), ptr(dop), ptr(dlogit), 0.3, 1.0,
                                          0.2, 2.0, ptr(dx), ptr(dl), None, None, None, ptr(losses), ptr(scratch),
                                          ptr(hdr[1:2]), C.byref(tail) if tail is not None else None, _stream()),
              "ggs_registration_aux_tail")
        torch.cuda.synchronize()
        return dx, dl, dlogit, losses

    ref = run([], 0, 0)
    host = torch.zeros(12, dtype=torch.float32).pin_memory()
    mapped = host_mapped_pointer(host)
    states = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(3)]
    expect = [s.clone() for s in states]
    for rep, guard in enumerate((0, 0, 1, 0)):
        got = run(states, mapped, guard)
        for a, b in zip(got, ref):                       # (the hinge sums are float atomics: equal up to their order)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        gp = hdr[1:2].data_ptr()
        check(L.ggs_adam_tick_multi(3, (C.c_void_p * 3)(*[t.data_ptr() for t in expect]), 0.9, 0.999, gp, _stream()), "tick")
        torch.cuda.synchronize()
        for a, b in zip(states, expect):
            assert torch.equal(a, b)
        assert host[0] == 12.5 and host[1] == 99.0 and torch.equal(host[2:5], got[3].cpu()) and not bool(host[5:8].any())
        assert host[8:12].view(torch.int64).tolist() == [4242, guard]
    assert int(states[0][:8].view(torch.int64)) == 3
    # a device block works the same way
    dev_block = torch.zeros(12, device="cuda")
    got = run([], dev_block.data_ptr(), 0)
    assert torch.equal(dev_block[2:5], got[3]) and torch.equal(dev_block[:2], sums) and dev_block[8:12].view(torch.int64).tolist() == [4242, 0]
