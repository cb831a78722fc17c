"""The measurement aids of round 6 through the C ABI: device timestamps captured inside a replayed graph (ggs_profile_stamps /
ggsplat.profile.DeviceStamps) and the evaluated-against-blended pair census of the two compositing kernels (ggs_count_pairs,
ggs_count_forward_visits).  Neither is on a product path; bench.py's roofline.in_graph / roofline.work are built on them."""
import ctypes as C

import pytest
import torch

from helpers import rel_l1, small_scene
from ggsplat import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(V=3, P=1500, W=144, H=96):
    from ggsplat import rasterizer as R
    sc, _ = small_scene(P=P, W=W, H=H, sh_degree=1, seed=12, scale_mul=5.0)
    cams = S.stack_cameras(S.orbit_cameras(V, width=W, img_height=H, fx=1.1 * W, fy=1.1 * W, cx=W / 2 - 1.0, cy=H / 2 + 2.0), device=DEV)
    t = {k: sc[k].to(DEV) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    kw = dict(view=cams["view"], proj=cams["proj"], campos=cams["campos"], tanfov=cams["tanfov"], bg=torch.zeros(3, device=DEV),
              W=W, H=H, sh_degree=1)
    args = (t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
    return R, args, kw


def test_pair_census_of_both_compositing_kernels():
    from ggsplat import _lib
    R, args, kw = _scene()
    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    *_, st = R.forward_views(*args, **kw)                       # learns the capacity; reference values
    one = torch.zeros(1, dtype=torch.int64, device=DEV)
    _lib.check(L.ggs_count_blends(C.byref(st.prm), st.geom.data_ptr(), st.bin.data_ptr(), st.cap, st.img.data_ptr(), one.data_ptr(), stream),
               "ggs_count_blends")
    sf = R.StagedForward(*args, **kw)
    c3, c4 = torch.full((3,), -1, dtype=torch.int64, device=DEV), torch.full((4,), -1, dtype=torch.int64, device=DEV)
    sf.run(sf.COUNT | sf.BIN)
    s2 = sf.state
    ids_before = R.bin_sections(s2)["ids"][:st.num_rendered].clone()
    _lib.check(L.ggs_count_forward_visits(C.byref(s2.prm), s2.geom.data_ptr(), s2.bin.data_ptr(), s2.cap, c3.data_ptr(), stream),
               "ggs_count_forward_visits")
    assert torch.equal(R.bin_sections(s2)["ids"][:st.num_rendered], ids_before)        # the census stores nothing
    sf.run(sf.COMPOSITE)
    _lib.check(L.ggs_count_pairs(C.byref(s2.prm), s2.geom.data_ptr(), s2.bin.data_ptr(), s2.cap, s2.img.data_ptr(), c4.data_ptr(), stream),
               "ggs_count_pairs")
    torch.cuda.synchronize()
    fwd_pass, fwd_blend_pass, fwd_entries = c3.tolist()
    blended, bwd_pass, bwd_reduced, bwd_walked = c4.tolist()
    assert blended == int(one.item()) > 0
    n = int(sf.header[0])
    assert n == st.num_rendered
    # forward: every entry walked is a list entry; a pass that blends is a pass; every blended pair lies in a blending pass of the
    # forward and in a pass of the backward.  (The backward's passes are NOT a subset of the forward's blending passes: a quadrant that
    # finished early keeps the bits of the entries behind it, and the backward walks to the last contributor of the whole tile.)
    assert 0 < fwd_entries <= n and fwd_blend_pass <= fwd_pass <= 4 * fwd_entries
    assert bwd_reduced <= bwd_walked <= n and 0 < bwd_reduced <= bwd_pass <= 4 * bwd_reduced
    assert blended <= 64 * bwd_pass and blended <= 64 * fwd_blend_pass
    # NULL arguments are argument errors
    assert L.ggs_count_pairs(C.byref(s2.prm), None, s2.bin.data_ptr(), s2.cap, s2.img.data_ptr(), c4.data_ptr(), stream) != 0
    assert L.ggs_count_forward_visits(C.byref(s2.prm), s2.geom.data_ptr(), s2.bin.data_ptr(), s2.cap, None, stream) != 0


def test_device_stamps_eager_and_inside_a_replayed_graph():
    from ggsplat.profile import DeviceStamps, NAMES
    R, args, kw = _scene()
    w = torch.randn(3, 3, 96, 144, generator=torch.Generator().manual_seed(2)).to(DEV)

    def step():
        *_, st = R.forward_views(*args, **kw)
        return R.backward_views(st, w)
    ref = step()
    torch.cuda.synchronize()
    # eager: the brackets pair up, every kernel of the forward and the backward has a positive interval
    st = DeviceStamps(DEV, capacity=256).start()
    g1 = step()
    st.stop()
    torch.cuda.synchronize()
    r = st.read()
    assert r["n_stamps"] % 2 == 0 and r["n_stamps"] >= 16 and r["clock_khz"] > 1000
    for k in ("preprocess", "scan_tiles", "scatter", "sort_tiles", "render_fwd", "render_bwd", "preprocess_bwd", "zero_fill"):
        assert r["launches"][k] >= 1 and 0.0 < r["seconds"][k] < 0.05, k
    assert 0.0 < sum(r["seconds"].values()) <= r["span"] and r["between_brackets"] >= 0.0
    for k in ref:
        assert rel_l1(g1[k], ref[k]) <= 1e-5, k           # the stamps change nothing (float atomics: summation order only)
    # after stop() nothing is stamped any more
    before = st.slots.clone()
    step()
    torch.cuda.synchronize()
    assert torch.equal(st.slots, before)
    # captured: the stamps are replayed with the step and refresh their slots on every replay
    st = DeviceStamps(DEV, capacity=256).start()
    step()                                        # (an eager call in front: restart() forgets its stamps)
    st.restart()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    st.stop()
    n_captured = len(st.ids)
    assert n_captured == r["n_stamps"]
    g.replay(); torch.cuda.synchronize()
    r1, s1 = st.read(), st.slots[:n_captured].clone()
    g.replay(); torch.cuda.synchronize()
    r2, s2 = st.read(), st.slots[:n_captured].clone()
    assert bool((s2 > s1).all()) and bool((s1[1:] >= s1[:-1]).all())            # one stream: the clock only moves forward
    for rr in (r1, r2):
        assert rr["launches"]["render_bwd"] == 1 and 0.0 < rr["seconds"]["render_bwd"] < 0.05
        assert abs(sum(rr["seconds"].values()) + rr["between_brackets"] - rr["span"]) < 1e-9
    for k in ref:
        assert rel_l1(out[k], ref[k]) <= 1e-5, k
    # capacity too small: later stamps are dropped, the log says how many there were
    st = DeviceStamps(DEV, capacity=4).start()
    step()
    st.stop()
    assert len(st.ids) == 4 and st.dropped == r["n_stamps"] - 4
    assert [NAMES[i // 2] for i in st.ids[:2]] == ["zero_fill", "zero_fill"]
