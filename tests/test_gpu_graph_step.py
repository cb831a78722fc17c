"""Graph-captured inner step (the caller of the hot path): GraphAdam against torch.optim.Adam, the
overflow guard, and GraphedRegistrationStep (one hipGraph replay per s2 iteration) against the eager
registration_step with torch's Adam on the same cameras -- parameters, moments-driven updates, densification
statistics and loss terms after several iterations."""
from types import SimpleNamespace

import math

import pytest
import torch

from ggsplat import synthetic as S

pytestmark = pytest.mark.gpu
W, H = 96, 80


def test_graph_adam_matches_torch_adam():
    from ggsplat.adam import GraphAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(1000, 3), (1000, 1, 3), (777,), (5, 4)]
    a = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    lrs = [1.6e-4, 2.5e-3, 5e-2, 1e-3]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    opt = GraphAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=1e-15)
    for it in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).cuda() * (10.0 ** (it - 3))
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if it == 3:                                   # schedule update between steps
            ref.param_groups[0]["lr"] = opt.param_groups[0]["lr"] = 7e-5
            opt.push_lr()
        ref.step()
        opt.step()
    assert opt.step_count == 6
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7), float((pa - pb).abs().max())
        for k in ("exp_avg", "exp_avg_sq"):
            x, y = opt.state[pa][k], ref.state[pb][k]
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-6 * float(y.abs().max())), (k, float((x - y).abs().max()))


def test_graph_adam_guard_voids_the_step():
    from ggsplat.adam import GraphAdam
    p = torch.randn(1001).cuda().requires_grad_(True)
    p.grad = torch.randn(1001).cuda()
    before = p.detach().clone()
    opt = GraphAdam([{"params": [p], "lr": 1e-2}])
    guard = torch.ones(1, dtype=torch.int64, device="cuda")
    opt.step(guard=guard)
    assert torch.equal(p.detach(), before) and opt.step_count == 0
    assert float(opt.state[p]["exp_avg"].abs().max()) == 0.0
    guard.zero_()
    opt.step(guard=guard)
    assert opt.step_count == 1 and not torch.equal(p.detach(), before)


def _scene(seed=0):
    v, f = S.skirt_mesh(24, 40, r_top=0.30, r_bottom=0.5, height=0.8, jitter=2e-3, seed=seed)
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=0, seed=seed)
    cams = S.rig_cameras(n_rings=2, n_az=3, radius=2.2, width=W, height=H, f=70.0, seed=seed)
    for cam in cams:
        for n in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(cam, n, getattr(cam, n).cuda())
    g = torch.Generator().manual_seed(seed + 9)
    gts = [torch.rand(3, H, W, generator=g).cuda() for _ in cams]
    masks = [(torch.rand(1, H, W, generator=g) > 0.2).float().cuda() for _ in cams]
    return v, f, params, cams, gts, masks


def _model(v, f, params, opt, graph: bool):
    from ggsplat.adam import GraphAdam
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
    m.training_setup(opt, is_ff=True, optimizer="torch")
    if graph:
        m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
    return m


@pytest.mark.parametrize("slack,lean", [(1.0, True), (0.002, True), (1.0, False), (0.002, False)])
def test_graphed_registration_step_matches_eager(slack, lean):
    """slack 0.002: the first capture runs with a binning capacity far too small -> the replay overflows, the guarded
    kernels change nothing, the step re-captures and replays; the trajectory must be the same."""
    from ggsplat import rasterizer as R
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep, registration_step
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.3, "threshold_scale": 0.02})
    v, f, params, cams, gts, masks = _scene()
    bg = torch.zeros(3, device="cuda")
    eager, graphed = _model(v, f, params, opt, False), _model(v, f, params, opt, True)
    R._cap_hint.clear()
    step = GraphedRegistrationStep(graphed, W, H, bg, opt=opt, capacity_slack=slack, lean=lean)
    names = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"]
    order = [0, 3, 1, 5, 2, 4, 0]
    for it, ci in enumerate(order):
        if it == 4:                                   # learning-rate schedule moves between iterations
            for m in (eager, graphed):
                m.optimizer.param_groups[0]["lr"] *= 0.5
            graphed.optimizer.push_lr()
        ref = registration_step(eager, cams[ci], gts[ci], masks[ci], bg, opt=opt, fused_loss=True)
        out = step(cams[ci], gts[ci], masks[ci])
        for k in ("img", "ssim", "xyz", "scale", "loss"):
            r = float(ref[k].detach())
            assert abs(float(out[k]) - r) <= 1e-4 * max(1.0, abs(r)), (it, k)
    assert step.recaptures == (0 if slack == 1.0 else 1)
    assert graphed.optimizer.step_count == len(order)
    # Both sides run the same kernels; the only difference is the order of the float atomics.  Adam with eps 1e-15
    # turns a gradient whose sign is decided by that rounding noise into a full +-lr step, so a handful of elements
    # may legitimately differ: require 99.5 % of every tensor within tolerance and a tiny mean deviation.
    def close(a, b, rtol, atol, what):
        ok = (a - b).abs() <= atol + rtol * b.abs()
        assert float(ok.float().mean()) >= 0.995, (what, float(ok.float().mean()), float((a - b).abs().max()))
        assert float((a - b).abs().mean()) <= 1e-5, (what, float((a - b).abs().mean()))

    for n in names:
        close(getattr(graphed, n).detach(), getattr(eager, n).detach(), 1e-4, 2e-6, n)
    close(graphed.mesh.v.detach(), eager.mesh.v.detach(), 1e-4, 2e-6, "mesh.v")
    close(graphed.xyz_gradient_accum, eager.xyz_gradient_accum, 1e-3, 1e-7, "xyz_gradient_accum")
    assert torch.equal(graphed.denom, eager.denom)
    assert torch.equal(graphed.max_radii2D, eager.max_radii2D)


def test_lean_step_without_mapped_pinned_memory_copies_its_blocks(monkeypatch):
    """The lean step reads its parameter block and writes its result block in place in pinned host memory when that memory is
    mapped into the device's address space; where it is not, one H2D and one D2H copy per iteration stand in.  Same numbers."""
    from ggsplat import _lib, rasterizer as R
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.3, "threshold_scale": 0.02})
    v, f, params, cams, gts, masks = _scene(seed=3)
    bg = torch.zeros(3, device="cuda")
    a, b = _model(v, f, params, opt, True), _model(v, f, params, opt, True)
    R._cap_hint.clear()
    mapped = GraphedRegistrationStep(a, W, H, bg, opt=opt)
    assert mapped._blk_map and mapped._out_map
    monkeypatch.setattr(_lib, "host_mapped_pointer", lambda t: 0)
    copied = GraphedRegistrationStep(b, W, H, bg, opt=opt)
    assert copied._blk_map == 0 and copied._out_map == 0
    for ci in (0, 2, 1, 3):
        o1, o2 = mapped(cams[ci], gts[ci], masks[ci]), copied(cams[ci], gts[ci], masks[ci])
        for k in ("img", "ssim", "xyz", "scale", "loss", "n_visible"):
            assert abs(o1[k] - o2[k]) <= 1e-5 * max(1.0, abs(o2[k])), (ci, k, o1[k], o2[k])
    assert a.optimizer.step_count == b.optimizer.step_count == 4


@pytest.mark.parametrize("lean", [True, False])
def test_graphed_mesh_only_step_without_mask(lean):
    """Later frames of a sequence: only mesh.v is optimised (training_setup(is_ff=False)), no hinge terms, no
    densification statistics; here also without a foreground mask."""
    from ggsplat import rasterizer as R
    from ggsplat.adam import GraphAdam
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep, registration_step
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "only_foreground_loss": False})
    v, f, params, cams, gts, masks = _scene(seed=2)
    bg = torch.ones(3, device="cuda")
    models = []
    for graph in (False, True):
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        m.training_setup(opt, is_ff=False, optimizer="torch")
        m.optimizer.param_groups[0]["lr"] = 1e-3
        if graph:
            m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
        models.append(m)
    eager, graphed = models
    R._cap_hint.clear()
    step = GraphedRegistrationStep(graphed, W, H, bg, opt=opt, first_frame_template=False, use_mask=False, lean=lean)
    for ci in (0, 2, 4, 1):
        ref = registration_step(eager, cams[ci], gts[ci], None, bg, opt=opt, first_frame_template=False, fused_loss=True)
        out = step(cams[ci], gts[ci], None)
        for k in ("img", "ssim", "loss"):
            r = float(ref[k].detach())
            assert abs(float(out[k]) - r) <= 1e-4 * max(1.0, abs(r)), (ci, k)
    a, b = graphed.mesh.v.detach(), eager.mesh.v.detach()
    ok = (a - b).abs() <= 2e-6 + 1e-4 * b.abs()
    assert float(ok.float().mean()) >= 0.995 and float((a - b).abs().mean()) <= 1e-5
    assert float((a - torch.as_tensor(v, device="cuda")).abs().max()) > 0      # the vertices did move
    assert torch.equal(graphed._xyz.detach(), eager._xyz.detach())              # nothing else was touched
    assert float(graphed.denom.abs().max()) == 0.0


def test_graph_adam_many_tensors_and_odd_sizes():
    """More than 16 tensors (ggs_adam_step_multi takes 16 per call), sizes that are not multiples of 4, an empty tensor
    and a tensor without gradient (skipped like torch does)."""
    from ggsplat.adam import GraphAdam
    g = torch.Generator().manual_seed(5)
    sizes = [1, 2, 3, 5, 7, 64, 255, 257, 1000, 1023, 4096, 33, 0, 9, 17, 31, 63, 65, 127, 129, 511]
    a = [torch.randn(n, generator=g).cuda().requires_grad_(True) for n in sizes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    ref = torch.optim.Adam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(b)], lr=0.0, eps=1e-15)
    opt = GraphAdam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(a)], lr=0.0, eps=1e-15)
    for it in range(3):
        for i, (pa, pb) in enumerate(zip(a, b)):
            if i == 5 and it == 1:
                pa.grad, pb.grad = None, None             # no gradient this step
                continue
            gr = torch.randn(pa.shape, generator=g).cuda()
            pa.grad, pb.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7)


def test_graphed_appearance_step_matches_eager_gather_semantics():
    """s3 iteration: a toy 'net' (learnable per-Gaussian offsets + a fixed visibility mask).  Eager side = the reference's
    semantics (boolean gather of the visible Gaussians, torch Adam); graphed side = one hipGraph replay per iteration with
    the mask applied to the opacities."""
    from ggsplat import rasterizer as R
    from ggsplat.adam import GraphAdam
    from ggsplat.inner_step import DEFAULT_OPT, GraphedAppearanceStep, appearance_step
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.05, "threshold_scale": 0.02})
    v, f, params, cams, gts, masks = _scene(seed=4)
    bg = torch.zeros(3, device="cuda")
    P = f.shape[0]
    g = torch.Generator().manual_seed(11)
    vis = (torch.rand(P, generator=g) > 0.4).cuda()

    class ToyNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            gg = torch.Generator().manual_seed(12)
            self.xyz_off = torch.nn.Parameter((torch.randn(P, 3, generator=gg) * 0.02).cuda())
            self.sh_off = torch.nn.Parameter((torch.randn(P, 1, 3, generator=gg) * 0.05).cuda())

        def forward(self, gaussians, cam):
            # the GPU ray cast (ggs_visibility) in place of the reference's open3d scene: no host round trip, so it
            # can live inside the captured step; AND-ed with a fixed random mask to get a ~40 % visible set
            from ggsplat.mesh_gaussian_model import visible_mask
            with torch.no_grad():
                seen = visible_mask(gaussians.mesh.v, gaussians.mesh.f, gaussians.binding,
                                    gaussians.get_anchor_points(), cam.camera_center)
            return self.xyz_off, self.sh_off, vis & seen

    sides = []
    for graph in (False, True):
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        net = ToyNet()
        groups = [{"params": [net.xyz_off], "lr": 1e-3, "name": "net_xyz"}, {"params": [net.sh_off], "lr": 5e-3, "name": "net_sh"},
                  {"params": [m._opacity], "lr": 1e-2, "name": "opacity"}, {"params": [m._scaling], "lr": 2e-3, "name": "scaling"}]
        o = GraphAdam(groups, lr=0.0, eps=1e-15) if graph else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        sides.append((m, net, o))
    (me, ne, oe), (mg, ng, og) = sides
    R._cap_hint.clear()
    step = GraphedAppearanceStep(mg, ng, W, H, bg, og, opt=opt)
    for ci in (0, 3, 1, 5, 2):
        ref = appearance_step(me, ne, cams[ci], gts[ci], masks[ci], bg, optimizer=oe, opt=opt, fused_loss=True)
        out = step(cams[ci], gts[ci], masks[ci])
        for k in ("img", "ssim", "xyz", "scale", "opacity", "loss"):
            r = float(ref[k].detach())
            assert abs(float(out[k]) - r) <= 1e-4 * max(1.0, abs(r)), (ci, k)
    assert step.recaptures == 0 and og.step_count == 5

    def close(a, b, what):
        ok = (a - b).abs() <= 2e-6 + 1e-4 * b.abs()
        assert float(ok.float().mean()) >= 0.995, (what, float(ok.float().mean()))
        assert float((a - b).abs().mean()) <= 1e-5, what
    close(ng.xyz_off.detach(), ne.xyz_off.detach(), "net.xyz_off")
    close(ng.sh_off.detach(), ne.sh_off.detach(), "net.sh_off")
    close(mg._opacity.detach(), me._opacity.detach(), "_opacity")
    close(mg._scaling.detach(), me._scaling.detach(), "_scaling")


def test_graphed_step_takes_host_images_and_follows_camera_edits():
    """What changes per iteration reaches a replay through ONE small device block: camera matrices (packed once per camera
    object) and the device addresses of ground truth / mask.  (1) Images that live on the HOST go through the landing buffers
    and give the same losses as the same images on the GPU; (2) a camera whose matrices are edited IN PLACE is re-packed (the
    cache follows the tensors' version counters), so the next replay renders the new view."""
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
    v, f, params, cams, gts, masks = _scene(3)
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "position_lr_init": 0.0, "feature_lr": 0.0, "opacity_lr": 0.0,
                             "scaling_lr": 0.0, "rotation_lr": 0.0})      # frozen parameters: the losses depend on the inputs only
    bg = torch.zeros(3, device="cuda")
    m = _model(v, f, params, opt, graph=True)
    for grp in m.optimizer.param_groups:
        grp["lr"] = 0.0
    m.optimizer.push_lr()
    step = GraphedRegistrationStep(m, W, H, bg, opt=opt)
    on_gpu = step(cams[0], gts[0], masks[0])
    on_host = step(cams[0], gts[0].cpu(), masks[0].cpu())
    assert abs(on_gpu["loss"] - on_host["loss"]) <= 1e-6 * abs(on_gpu["loss"])
    other = step(cams[1], gts[0], masks[0])
    assert abs(other["img"] - on_gpu["img"]) > 1e-4                      # another camera, another image
    saved = [getattr(cams[0], n).clone() for n in ("world_view_transform", "full_proj_transform", "camera_center")]
    for n in ("world_view_transform", "full_proj_transform", "camera_center"):
        getattr(cams[0], n).copy_(getattr(cams[1], n))                    # in place: same tensor objects, new versions
    cams[0].FoVx, cams[0].FoVy = cams[1].FoVx, cams[1].FoVy
    edited = step(cams[0], gts[0], masks[0])
    assert abs(edited["loss"] - other["loss"]) <= 1e-6 * abs(other["loss"])
    for n, t in zip(("world_view_transform", "full_proj_transform", "camera_center"), saved):
        getattr(cams[0], n).copy_(t)
    assert step.recaptures == 0


def test_graphed_appearance_step_with_a_convolutional_net():
    """The captured s3 iteration with a NETWORK producing the offsets (what bench.py's `config4_s3_with_network` line times):
    ggsplat.stylenet.StyleUNetLite on the HIP fused_bias_act / upfirdn2d ops + PyTorch convolutions inside the hipGraph, sampled
    at per-Gaussian UV coordinates.  Replays against the eager appearance_step with torch.optim.Adam on a twin."""
    import copy
    from ggsplat.adam import GraphAdam
    from ggsplat.inner_step import DEFAULT_OPT, GraphedAppearanceStep, appearance_step
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.stylenet import CHANNELS, StyleUNetLite, TexelOffsets
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.05, "threshold_scale": 0.02})
    v, f, params, cams, gts, masks = _scene(seed=6)
    P = f.shape[0]
    bg = torch.zeros(3, device="cuda")
    g = torch.Generator().manual_seed(21)
    uv, vis = torch.rand(P, 2, generator=g).cuda(), (torch.rand(P, generator=g) > 0.3).cuda()
    cond = torch.randn(1, 4, 64, 64, generator=g).cuda()
    torch.manual_seed(22)
    unet = StyleUNetLite(size=64, in_ch=4, out_ch=6, style_dim=32, impl="hip", channels={k: min(c, 16) for k, c in CHANNELS.items()})
    net_g = TexelOffsets(unet, uv, 1, vis, cond).cuda()
    net_e = copy.deepcopy(net_g)
    sides = []
    for net, graph in ((net_e, False), (net_g, True)):
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        groups = [{"params": list(net.parameters()), "lr": 1e-3, "name": "net"}, {"params": [m._opacity], "lr": 1e-2, "name": "opacity"}]
        o = GraphAdam(groups, lr=0.0, eps=1e-15) if graph else torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        sides.append((m, o))
    (me, oe), (mg, og) = sides
    step = GraphedAppearanceStep(mg, net_g, W, H, bg, og, opt=opt)
    for ci in (0, 2, 4):
        ref = appearance_step(me, net_e, cams[ci], gts[ci], masks[ci], bg, optimizer=oe, opt=opt, fused_loss=True)
        out = step(cams[ci], gts[ci], masks[ci])
        for k in ("img", "ssim", "loss"):
            r = float(ref[k].detach())
            assert abs(float(out[k]) - r) <= 2e-4 * max(1.0, abs(r)), (ci, k, float(out[k]), r)
    assert step.recaptures == 0 and og.step_count == 3
    # the two nets trained in step: the style mapping's first layer after three Adam steps
    wg, we = net_g.net.mapping[0].weight.detach(), net_e.net.mapping[0].weight.detach()
    assert float((wg - we).abs().mean()) <= 5e-4 * float(we.abs().mean()) + 1e-7


def test_training_setup_default_is_graph_adam_with_state_dict_round_trip():
    """training_setup() on the GPU builds a GraphAdam (one launch for all parameter tensors); its state_dict() has torch's
    layout and load_state_dict() restores moments, step counts and bias corrections: a restored optimiser continues exactly
    like the original, and like torch.optim.Adam within 2e-6."""
    from ggsplat.adam import GraphAdam
    from ggsplat.inner_step import DEFAULT_OPT
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f, params, cams, gts, masks = _scene(seed=3)
    gen = torch.Generator(device="cuda").manual_seed(5)

    def make(kind):
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        m.training_setup(DEFAULT_OPT, is_ff=True, optimizer=kind)
        return m
    a, b, t = make("auto"), make("auto"), make("torch")
    assert isinstance(a.optimizer, GraphAdam) and isinstance(t.optimizer, torch.optim.Adam)
    assert [g["name"] for g in a.optimizer.param_groups] == [g["name"] for g in t.optimizer.param_groups]
    grads = [[torch.randn(p.shape, device="cuda", generator=gen) * 1e-2 for p in a.parameters()] for _ in range(5)]

    def run(m, gs):
        for g in gs:
            for p, gi in zip(m.parameters(), g):
                p.grad = gi.clone()
            m.optimizer.step()
            m.optimizer.zero_grad()
    run(a, grads[:3]); run(t, grads)
    sd = a.optimizer.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 7 and float(sd["state"][0]["step"]) == 3.0
    with torch.no_grad():
        for pb, pa in zip(b.parameters(), a.parameters()):
            pb.copy_(pa)
    b.optimizer.load_state_dict(sd)
    run(a, grads[3:]); run(b, grads[3:])
    for pa, pb, pt in zip(a.parameters(), b.parameters(), t.parameters()):
        pa, pb, pt = pa.detach(), pb.detach(), pt.detach()
        assert torch.equal(pa, pb)
        if pa.numel():                              # _features_rest is empty at SH degree 0
            assert float((pa - pt).abs().max()) <= 2e-6 * float(pt.abs().max()) + 1e-9


def test_graph_adam_and_torch_adam_exchange_state_dicts():
    """a checkpoint written with one optimiser class loads into the other and training continues the same way
    (torch.optim.Adam -> GraphAdam, GraphAdam -> torch.optim.Adam), within the 2e-6 the two updates agree to."""
    from ggsplat.adam import GraphAdam
    gen = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(257, 3), (64, 1, 3), (1000,)]
    init = [torch.randn(s, device="cuda", generator=gen) for s in shapes]
    grads = [[torch.randn(s, device="cuda", generator=gen) * 1e-2 for s in shapes] for _ in range(6)]

    def make(kind):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        groups = [{"params": [p], "lr": 1e-3 * (i + 1), "name": f"g{i}"} for i, p in enumerate(ps)]
        return ps, (GraphAdam(groups, lr=0.0, eps=1e-15) if kind == "graph" else torch.optim.Adam(groups, lr=0.0, eps=1e-15))

    def run(ps, opt, gs):
        for g in gs:
            for p, gi in zip(ps, g):
                p.grad = gi.clone()
            opt.step()
            opt.zero_grad()
    ref_p, ref_o = make("torch")
    run(ref_p, ref_o, grads)                                   # six torch steps: the reference trajectory
    for first, second in (("torch", "graph"), ("graph", "torch")):
        pa, oa = make(first)
        run(pa, oa, grads[:3])
        pb, ob = make(second)
        with torch.no_grad():
            for b, a in zip(pb, pa):
                b.copy_(a)
        ob.load_state_dict(oa.state_dict())
        run(pb, ob, grads[3:])
        for b, r in zip(pb, ref_p):
            assert float((b.detach() - r.detach()).abs().max()) <= 2e-6 * float(r.detach().abs().max()) + 1e-9, (first, second)
    bad = ref_o.state_dict()
    bad["param_groups"][0]["amsgrad"] = True
    with pytest.raises(ValueError, match="amsgrad"):
        make("graph")[1].load_state_dict(bad)


@pytest.mark.parametrize("slack", [1.0, 0.002])
def test_pipelined_registration_step_matches_the_sequential_replay(slack):
    """PipelinedRegistrationStep: two captured copies of the lean iteration replayed alternately, results read one iteration
    late.  Same kernels in the same order on one stream as GraphedRegistrationStep -> the same losses (shifted by one call) and
    the same parameters after N iterations, up to the order of the float atomics of the render backward.  slack 0.002: each
    capture runs with a binning capacity far too small (as in test_graphed_registration_step_matches_eager), every view overflows its first replay -> the recovery path (re-capture, re-run)."""
    from ggsplat.inner_step import GraphedRegistrationStep, PipelinedRegistrationStep
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.inner_step import DEFAULT_OPT
    v, f, params, cams, gts, masks = _scene(seed=2)
    W_, H_ = cams[0].image_width, cams[0].image_height
    bg = torch.zeros(3, device="cuda")

    def model():
        m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
        m.training_setup(DEFAULT_OPT, is_ff=True)
        return m
    from ggsplat import rasterizer as R
    a, b = model(), model()
    R._cap_hint.clear()
    seq = GraphedRegistrationStep(a, W_, H_, bg)
    pip = PipelinedRegistrationStep(b, W_, H_, bg, capacity_slack=slack)
    n = 7
    ref, got = [], []
    for i in range(n):
        c, gt, mk = cams[i % len(cams)], gts[i % len(cams)], masks[i % len(cams)]
        ref.append(seq(c, gt, mk))
        r = pip(c, gt, mk)
        if i == 0:
            assert r is None
        else:
            got.append(r)
    got.append(pip.flush())
    assert pip.flush() is None and len(got) == n
    if slack == 1.0:
        assert pip.recaptures == 0
        for r, g_ in zip(ref, got):               # the same iteration, read one call later
            assert abs(r["loss"] - g_["loss"]) <= 1e-5 * abs(r["loss"]) + 1e-7
    else:
        assert pip.recaptures >= 2 and all(math.isfinite(g_["loss"]) for g_ in got)
    # Same kernels on both sides; the float atomics of the render backward add in another order, and Adam with eps 1e-15 turns
    # a gradient whose sign is decided by that noise into a full +-lr step (see test_graphed_registration_step_matches_eager):
    # 99.5 % of every tensor within tolerance.  With overflows an iteration is re-run behind its successor: only a small mean
    # deviation is required.
    for pa, pb in zip(a.parameters(), b.parameters()):
        if not pa.numel():
            continue
        x, y = pb.detach(), pa.detach()
        if slack == 1.0:
            ok = (x - y).abs() <= 2e-6 + 1e-4 * y.abs()
            assert float(ok.float().mean()) >= 0.995 and float((x - y).abs().mean()) <= 1e-5
        else:
            assert float((x - y).abs().mean()) <= 5e-3 and bool(torch.isfinite(x).all())


def test_replicas_follow_their_solo_trajectories():
    """ReplicaRegistrationSteps: three independent registrations (different parameter jitters, different camera orders), each with the
    reference's one-step-per-view semantics, replayed side by side on three streams.  Every replica must follow the trajectory of
    the SAME problem run alone through GraphedRegistrationStep: same kernels on the same inputs, so the two differ by the order of
    the float atomics in the render backward and by nothing else -- the bound is what two SOLO runs of one problem differ by
    (measured here), not bit identity, which the float atomics do not give a solo run either.  A capacity far too small for
    replica 1 exercises the overflow recovery beside replicas that keep running."""
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep, ReplicaRegistrationSteps
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f, params, cams, gts, masks = _scene(seed=4)
    W_, H_ = cams[0].image_width, cams[0].image_height
    bg = torch.zeros(3, device="cuda")
    R_, n = 3, 6

    def model(r):
        g = torch.Generator().manual_seed(100 + r)
        p = {k: (t + 0.01 * torch.randn(t.shape, generator=g) if k in ("_scaling", "_opacity") else t) for k, t in params.items()}
        m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=0, device="cuda")
        m.training_setup(DEFAULT_OPT, is_ff=True)
        return m

    def view(r, i):
        k = (i + 2 * r) % len(cams)
        return cams[k], gts[k], masks[k]

    def solo(r):
        m = model(r)
        st = GraphedRegistrationStep(m, W_, H_, bg)
        return m, [st(*view(r, i)) for i in range(n)]
    solos = [solo(r) for r in range(R_)]
    twin0, twin0_losses = solo(0)                       # the run-to-run noise of a solo run
    models = [model(r) for r in range(R_)]
    reps = ReplicaRegistrationSteps(models, W_, H_, bg)
    assert len(reps) == R_ and len({s.cuda_stream for s in reps.streams}) == R_
    losses = [[] for _ in range(R_)]
    for i in range(n):
        c, g, m = zip(*[view(r, i) for r in range(R_)])
        for r, out in enumerate(reps(c, g, m)):
            losses[r].append(out)
    reps.synchronize()
    assert reps.recaptures == 0

    def dist(ma, mb):
        return max(float((pa.detach() - pb.detach()).abs().mean()) for pa, pb in zip(ma.parameters(), mb.parameters()) if pa.numel())
    noise = dist(solos[0][0], twin0)
    for r in range(R_):
        for a, b in zip(solos[r][1], losses[r]):
            assert abs(a["loss"] - b["loss"]) <= 1e-5 * abs(a["loss"]) + 1e-7, r
        assert dist(solos[r][0], models[r]) <= max(4.0 * noise, 1e-5), (r, noise)
    # replicas are different problems: their parameters differ by far more than the noise
    assert dist(models[0], models[1]) > 100 * max(noise, 1e-7)
    # overflow recovery of ONE replica (capacity cut to 0.2 % at its capture) beside two that run normally
    from ggsplat import rasterizer as Rz
    models2 = [model(r) for r in range(R_)]
    reps2 = ReplicaRegistrationSteps(models2, W_, H_, bg)
    reps2.steps[1]._slack = 0.002
    for i in range(n):
        c, g, m = zip(*[view(r, i) for r in range(R_)])
        outs = reps2(c, g, m)
        assert all(math.isfinite(o["loss"]) for o in outs)
    assert reps2.steps[1].recaptures >= 1
    for r in (0, 2):
        assert dist(solos[r][0], models2[r]) <= max(4.0 * noise, 1e-5), r
    assert dist(solos[1][0], models2[1]) <= 5e-3


def _silhouettes(cams, models_v_f_params, bg):
    """garment silhouettes of the cameras: the initial model's own alpha > 0.05, as a segmentation mask would be"""
    from ggsplat.inner_step import DEFAULT_PIPE
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.render import render
    v, f, params = models_v_f_params
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
    out = []
    with torch.no_grad():
        m.update_face_coor()
        for c in cams:
            out.append((render(c, m, DEFAULT_PIPE, bg)["alpha"] > 0.05).float().reshape(1, c.image_height, c.image_width).contiguous())
    return out


@pytest.mark.parametrize("pipelined", [False, True])
def test_lean_step_with_the_sparse_mask_loss_gives_the_same_trajectory(pipelined):
    """sparse_mask=True (ggs_photometric_forward_sparse: the first loss pass skips the boxes without a mask pixel) against
    sparse_mask=False on silhouette masks: same losses, same parameters; sparse_mask=None picks the sparse form for a
    silhouette and the plain one for a dense mask."""
    from ggsplat import rasterizer as R
    from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep, PipelinedRegistrationStep
    opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.3, "threshold_scale": 0.02})
    v, f, params, cams, gts, dense_masks = _scene(seed=5)
    bg = torch.zeros(3, device="cuda")
    masks = _silhouettes(cams, (v, f, params), bg)
    frac = float(torch.stack(masks).mean())
    assert 0.01 < frac < 0.45, frac
    a, b, c = (_model(v, f, params, opt, True) for _ in range(3))
    R._cap_hint.clear()
    cls = PipelinedRegistrationStep if pipelined else GraphedRegistrationStep
    plain, sparse, auto = cls(a, W, H, bg, opt=opt, sparse_mask=False), cls(b, W, H, bg, opt=opt, sparse_mask=True), cls(c, W, H, bg, opt=opt)
    order = [0, 3, 1, 5, 2, 4, 0, 2]
    outs = {0: [], 1: [], 2: []}
    for ci in order:
        for k, st in enumerate((plain, sparse, auto)):
            outs[k].append(st(cams[ci], gts[ci], masks[ci]))
    if pipelined:
        for k, st in enumerate((plain, sparse, auto)):
            outs[k] = outs[k][1:] + [st.flush()]
    for o1, o2, o3 in zip(outs[0], outs[1], outs[2]):
        for k in ("img", "ssim", "xyz", "scale", "loss", "n_visible"):
            assert abs(o1[k] - o2[k]) <= 2e-5 * max(1.0, abs(o1[k])), (k, o1[k], o2[k])
            assert abs(o1[k] - o3[k]) <= 2e-5 * max(1.0, abs(o1[k])), (k, o1[k], o3[k])
    firsts = [st.steps[0] if pipelined else st for st in (plain, sparse, auto)]
    assert [s._sparse for s in firsts] == [False, True, True]
    for pa, pb in zip(a.parameters(), b.parameters()):
        if pa.numel():
            ok = (pa.detach() - pb.detach()).abs() <= 2e-6 + 1e-4 * pa.detach().abs()
            assert float(ok.float().mean()) >= 0.995 and float((pa.detach() - pb.detach()).abs().mean()) <= 1e-5
    # masks uploaded afresh every iteration are not kept alive by the tile-table cache (weak references)
    import gc
    for _ in range(6):
        sparse(cams[0], gts[0], masks[0].clone())
    if pipelined:
        sparse.flush()
    gc.collect()
    alive = sum(e[0]() is not None for e in firsts[1]._mask_tiles.values())
    assert alive <= len(set(order)) + 2, alive
    # a dense mask settles sparse_mask=None the other way
    d = GraphedRegistrationStep(_model(v, f, params, opt, True), W, H, bg, opt=opt)
    d(cams[0], gts[0], dense_masks[0])
    assert d._sparse is False
