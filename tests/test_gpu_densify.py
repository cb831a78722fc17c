"""Adaptive density control of the mesh-bound model (ggsplat.densify) against RECORDINGS OF THE REFERENCE'S OWN functions
(scene/mesh_gaussian_model.py:130-208, scene/gaussian_model.py:276-412, imported and run by tests/golden/make_golden.py ->
tests/golden/densify.npz) on a 1000-Gaussian fixture, with torch.optim.Adam and with GraphAdam, and the captured registration
step re-capturing itself when P changes."""
import copy
import math
import os

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from ggsplat import synthetic as S  # noqa: E402
from ggsplat.adam import GraphAdam  # noqa: E402
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel  # noqa: E402

DEV = "cuda"


def small_model(seed=0, graph_adam=False):
    v, f = S.skirt_mesh(n_around=40, n_rows=25)                 # 2000 faces
    p = S.skirt_gaussian_params(f.shape[0], sh_degree=1, seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    p["_xyz"] = torch.randn(f.shape[0], 3, generator=g) * 0.05
    m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=1, device=DEV)
    m.training_setup(DEFAULT_OPT, is_ff=True, optimizer="torch")
    if graph_adam:
        m.optimizer = GraphAdam(m.optimizer.param_groups, eps=1e-15)
    P = m._xyz.shape[0]
    gen = torch.Generator(device=DEV).manual_seed(seed + 9)
    # optimiser state with recognisable moments (one real step), then statistics that trigger clone / split / prune
    for prm in m.parameters():
        prm.grad = torch.randn(prm.shape, device=DEV, generator=gen) * 1e-3
    m.optimizer.step()
    m.optimizer.zero_grad()
    m.xyz_gradient_accum = torch.rand(P, 1, device=DEV, generator=gen) * 4e-4
    m.denom = torch.randint(0, 3, (P, 1), device=DEV, generator=gen).float()        # zeros -> NaN -> 0 like the reference
    m.max_radii2D = torch.rand(P, device=DEV, generator=gen) * 30
    with torch.no_grad():
        m._opacity[::17] = -8.0                                                       # below min_opacity: pruned
    return m


G = os.path.join(os.path.dirname(__file__), "golden", "densify.npz")
NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def golden_model(d, graph_adam=False):
    """The fixture of tests/golden/make_golden.py::model_golden on the GPU, brought to the state the reference's density
    control started from there: training_setup, one optimiser step on the golden gradients."""
    from types import SimpleNamespace
    t = np.load(os.path.join(os.path.dirname(G), "training_setup.npz"))
    opt = SimpleNamespace(**{str(k): float(v) for k, v in zip(t["opt_keys"], t["opt_vals"])})
    opt.position_lr_max_steps = int(opt.position_lr_max_steps)
    params = {k: torch.tensor(d["p" + k]) for k in NAMES}
    params["binding"] = torch.arange(d["faces"].shape[0])
    m = MeshGaussianModel.from_tensors(torch.tensor(d["verts"]), torch.tensor(d["faces"]), params, sh_degree=1, device=DEV)
    m.training_setup(opt, is_ff=True, optimizer="torch")
    if graph_adam:
        m.optimizer = GraphAdam(m.optimizer.param_groups, eps=1e-15)
    for n in NAMES:
        getattr(m, n).grad = torch.tensor(d["g_" + n]).to(DEV)
    m.mesh.v.grad = torch.tensor(d["g_vertex"]).to(DEV)
    m.optimizer.step()
    m.optimizer.zero_grad()
    m.update_face_coor()
    return m


def close(a, ref, rtol, atol):
    return torch.allclose(a.detach().cpu().float(), torch.as_tensor(ref).float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("graph_adam", [False, True])
@pytest.mark.parametrize("tag,max_screen_size", [("none", None), ("s20", 20)])
def test_densify_and_prune_matches_the_reference(graph_adam, tag, max_screen_size, monkeypatch):
    """ggsplat.densify against a RECORDING OF THE REFERENCE ITSELF: tests/golden/densify.npz holds the state after the
    reference's own training_setup / optimizer.step / densify_and_prune (scene/mesh_gaussian_model.py:130-208,
    scene/gaussian_model.py:276-408, imported from /root/reference by tests/golden/make_golden.py and run on the CPU).  The
    split's torch.normal draws are the recorded ones (mean + z * std on both sides).  Integers (P, binding, counters, which
    Gaussians were cloned / split / pruned) must agree exactly; floats to the rounding of CPU vs GPU exp / log / sigmoid /
    Adam and of the HIP binding kernel (the children of a split take their position and scale from it)."""
    d = np.load(G)
    m = golden_model(d, graph_adam)
    # state before density control: parameters and both Adam moments after one step on the golden gradients
    for n in NAMES:
        p_ = getattr(m, n)
        assert close(p_, d["pre" + n], 1e-5, 1e-7), n
        st = m.optimizer.state[p_]
        assert close(st["exp_avg"], d[f"pre{n}_m1"], 1e-5, 1e-12) and close(st["exp_avg_sq"], d[f"pre{n}_m2"], 1e-5, 1e-15), n
    assert close(m.mesh.v, d["pre_verts"], 1e-5, 1e-7)
    max_grad, min_opacity, extent = (float(x) for x in d["hyper"])
    m.xyz_gradient_accum = torch.tensor(d["s_accum"]).to(DEV)
    m.denom = torch.tensor(d["s_denom"]).to(DEV)
    m.max_radii2D = torch.tensor(d["s_radii"]).to(DEV)
    z = torch.tensor(d[f"{tag}_z"]).to(DEV)
    draws = []

    def recorded_normal(mean, std):
        draws.append(std.shape)
        assert tuple(std.shape) == tuple(z.shape), "the split selected other Gaussians than the reference's"
        return mean + z * std
    monkeypatch.setattr(torch, "normal", recorded_normal)
    P0 = m._xyz.shape[0]
    m.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)
    monkeypatch.undo()
    assert len(draws) == 1
    P1 = m._xyz.shape[0]
    assert P1 == d[f"{tag}_xyz"].shape[0] and P1 != P0
    assert torch.equal(m.binding.cpu(), torch.tensor(d[f"{tag}_binding"]))
    assert torch.equal(m.binding_counter.cpu(), torch.tensor(d[f"{tag}_counter"]))
    assert int(m.binding_counter.min()) >= 1                    # no face lost all its Gaussians
    for n in NAMES:
        got = getattr(m, n)
        assert isinstance(got, torch.nn.Parameter) and got.requires_grad and got.is_contiguous()
        # children of a split: world-frame sample R n + get_xyz and log(get_scaling / face_scaling / 1.6) -- through the HIP
        # binding kernel here, through the reference's bmm / roma-free arithmetic there
        tol = 3e-5 if n in ("_xyz", "_scaling") else 1e-5
        assert close(got, d[tag + n], tol, tol if n in ("_xyz", "_scaling") else 1e-7), n
        st = m.optimizer.state[got]
        assert close(st["exp_avg"], d[f"{tag}{n}_m1"], 1e-5, 1e-12) and close(st["exp_avg_sq"], d[f"{tag}{n}_m2"], 1e-5, 1e-15), n
        # moments of new Gaussians are exactly zero, as in the reference
        assert torch.equal(st["exp_avg"].cpu() == 0, torch.tensor(d[f"{tag}{n}_m1"]) == 0), n
    assert torch.equal(m.xyz_gradient_accum.cpu(), torch.tensor(d[f"{tag}_accum"]))
    assert torch.equal(m.denom.cpu(), torch.tensor(d[f"{tag}_denom"]))
    assert torch.equal(m.max_radii2D.cpu(), torch.tensor(d[f"{tag}_radii"]))
    # the optimiser drives the new tensors: a step changes them, and only them
    groups = {g["name"]: g["params"][0] for g in m.optimizer.param_groups}
    assert groups["xyz"] is m._xyz and groups["opacity"] is m._opacity and groups["vertex"] is m.mesh.v
    before = m._xyz.detach().clone()
    m._xyz.grad = torch.ones_like(m._xyz)
    m.optimizer.step()
    assert not torch.equal(before, m._xyz.detach())


def test_add_densification_stats_and_prune_rules_match_the_reference():
    """add_densification_stats (scene/gaussian_model.py:410-412) and prune_points / densify_and_clone on their own
    (scene/mesh_gaussian_model.py:130-156, :191-208) against the reference's recorded results: a request to prune EVERYTHING
    removes nothing (every face keeps its Gaussian), a clone of everything doubles every counter, and a request that takes
    both Gaussians of some faces and one of others is honoured exactly where the reference honours it."""
    d = np.load(G)
    m = golden_model(d)
    P = m._xyz.shape[0]
    from types import SimpleNamespace
    vsp = SimpleNamespace(grad=torch.tensor(d["ads_grad"]).to(DEV))
    filt = torch.tensor(d["ads_filter"]).to(DEV)
    m.add_densification_stats(vsp, filt)
    m.add_densification_stats(vsp, filt)
    assert close(m.xyz_gradient_accum, d["ads_accum"], 1e-6, 0.0) and torch.equal(m.denom.cpu(), torch.tensor(d["ads_denom"]))
    m.max_radii2D = torch.zeros(P, device=DEV)
    m.prune_points(torch.ones(P, dtype=torch.bool, device=DEV))
    assert m._xyz.shape[0] == int(d["pruneall_P"]) and torch.equal(m.binding_counter.cpu(), torch.tensor(d["pruneall_counter"]))
    m.percent_dense = 1e9
    m.densify_and_clone(torch.ones(P, 1, device=DEV), 0.5, 1.0)
    assert m._xyz.shape[0] == int(d["cloneall_P"]) and torch.equal(m.binding_counter.cpu(), torch.tensor(d["cloneall_counter"]))
    m.prune_points(torch.tensor(d["prunesome_mask"]).to(DEV))
    assert torch.equal(m.binding.cpu(), torch.tensor(d["prunesome_binding"]))
    assert torch.equal(m.binding_counter.cpu(), torch.tensor(d["prunesome_counter"]))
    assert close(m._xyz, d["prunesome_xyz"], 1e-5, 1e-7)


def test_prune_never_empties_a_face():
    m = small_model()
    P = m._xyz.shape[0]
    mask = torch.ones(P, dtype=torch.bool, device=DEV)          # ask for everything: every face keeps its Gaussian
    m.prune_points(mask)
    assert m._xyz.shape[0] == P and int(m.binding_counter.min()) == 1
    # clone everything once, then ask to prune one of the two Gaussians of every face
    grads = torch.ones(P, 1, device=DEV)
    m.percent_dense = 1e9
    m.densify_and_clone(grads, 0.5, 1.0)
    assert m._xyz.shape[0] == 2 * P and int(m.binding_counter.min()) == 2
    mask = torch.zeros(2 * P, dtype=torch.bool, device=DEV)
    mask[:P] = True
    m.prune_points(mask)
    assert m._xyz.shape[0] == P and torch.equal(m.binding_counter, torch.ones_like(m.binding_counter))


def test_captured_step_recaptures_when_density_control_changes_P():
    W, H = 256, 192
    m = small_model(graph_adam=True)
    cams = S.rig_cameras(n_rings=1, n_az=4, width=W, height=H, f=200.0, device=DEV)
    bg = torch.zeros(3, device=DEV)
    from ggsplat.render import render
    from ggsplat.inner_step import DEFAULT_PIPE
    with torch.no_grad():
        gts = [render(c, m, DEFAULT_PIPE, bg)["render"].clone() for c in cams]
    mask = torch.ones(1, H, W, device=DEV)
    step = GraphedRegistrationStep(m, W, H, bg)
    for i in range(4):
        d0 = step(cams[i % 4], gts[i % 4], mask)
    assert step.recaptures == 0 and math.isfinite(d0["loss"])
    P0 = m._xyz.shape[0]
    m.densify_and_prune(1e-7, 0.005, 1.0, None)
    P1 = m._xyz.shape[0]
    assert P1 != P0
    # eager twin of the model after the density step: the replayed (re-captured) iteration must match it
    twin = copy.deepcopy(m)

    def twin_param(name):
        return twin.mesh.v if name == "vertex" else getattr(twin, _attr(name))
    twin.optimizer = GraphAdam([{"name": g["name"], "lr": g["lr"], "params": [twin_param(g["name"])]}
                                for g in m.optimizer.param_groups], eps=1e-15)
    for g in m.optimizer.param_groups:                              # same moments / step counts
        src, dst = m.optimizer.state[g["params"][0]], twin.optimizer.state[twin_param(g["name"])]
        for k in src:
            dst[k].copy_(src[k])
    from ggsplat.inner_step import registration_step
    d1 = step(cams[0], gts[0], mask)
    assert step.recaptures == 1 and math.isfinite(d1["loss"])
    registration_step(twin, cams[0], gts[0].clone(), mask.clone(), bg, fused_loss=True)
    for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc"):
        a, b = getattr(m, n).detach(), getattr(twin, n).detach()
        assert a.shape[0] == P1
        assert float((a - b).abs().sum() / b.abs().sum().clamp_min(1e-12)) < 1e-4, n
    assert float((m.mesh.v.detach() - twin.mesh.v.detach()).abs().max()) < 1e-5


def _attr(name):
    return {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation"}[name]


def test_pipelined_step_survives_density_control_between_calls():
    """PipelinedRegistrationStep needs no flush() around densify_and_prune: the density step runs on the same stream behind the
    queued iteration, and each of the two captured copies re-captures itself on its next launch (P changed).  Same final state as
    the sequential captured step driven through the same calls."""
    from ggsplat.inner_step import DEFAULT_PIPE, PipelinedRegistrationStep
    from ggsplat.render import render
    W, H = 256, 192
    cams = S.rig_cameras(n_rings=1, n_az=4, width=W, height=H, f=200.0, device=DEV)
    bg = torch.zeros(3, device=DEV)
    a, b = small_model(graph_adam=True), small_model(graph_adam=True)
    with torch.no_grad():
        gts = [render(c, a, DEFAULT_PIPE, bg)["render"].clone() for c in cams]
    mask = torch.ones(1, H, W, device=DEV)
    seq, pip = GraphedRegistrationStep(a, W, H, bg), PipelinedRegistrationStep(b, W, H, bg)
    out_seq, out_pip = [], []
    for i in range(8):
        if i == 4:                                    # density control in the middle, with an iteration of `pip` still in flight
            torch.manual_seed(5); a.densify_and_prune(1e-7, 0.005, 1.0, None)
            torch.manual_seed(5); b.densify_and_prune(1e-7, 0.005, 1.0, None)
        out_seq.append(seq(cams[i % 4], gts[i % 4], mask))
        r = pip(cams[i % 4], gts[i % 4], mask)
        if r is not None:
            out_pip.append(r)
    out_pip.append(pip.flush())
    assert len(out_pip) == 8 and seq.recaptures == 1 and pip.recaptures == 2
    assert a._xyz.shape[0] == b._xyz.shape[0] != 2000
    for x, y in zip(out_seq, out_pip):
        assert math.isfinite(y["loss"]) and abs(x["loss"] - y["loss"]) <= 1e-3 * abs(x["loss"]) + 1e-6
    for n in ("_xyz", "_scaling", "_opacity", "_features_dc"):
        p, q = getattr(a, n).detach(), getattr(b, n).detach()
        assert float((p - q).abs().mean()) <= 1e-4, n


def test_getters_schedule_and_reset_opacity_match_the_reference():
    """The mesh-bound getters (get_xyz / get_scaling through the fused HIP binding kernel; scene/mesh_gaussian_model.py:105-128),
    get_opacity / get_features / get_covariance, update_learning_rate (scene/gaussian_model.py:171-177) and reset_opacity
    (:212-215 + replace_tensor_to_optimizer :261-274) of the model mirror against what the REFERENCE's own model class returned
    on the same state (tests/golden/densify.npz).  get_rotation needs roma, which the authoring image lacks: its expected value
    is composed from what the reference DID record on that state -- face_orien_mat, binding, _rotation -- in the order the
    reference's get_xyz fixes (:117-128): R(get_rotation) = face_orien_mat[binding] R(normalize(_rotation)), compared sign-free."""
    from helpers import quat_wxyz_to_rotmat
    d = np.load(G)
    m = golden_model(d)
    with torch.no_grad():
        assert close(m.face_orien_mat, d["face_orien_mat"], 2e-6, 2e-6)
        assert np.array_equal(m.binding.cpu().numpy(), d["getters_binding"]) and close(m._rotation, d["getters_rotation_raw"], 1e-6, 1e-7)
        want = torch.tensor(d["face_orien_mat"]).double()[torch.tensor(d["getters_binding"])] @ quat_wxyz_to_rotmat(d["getters_rotation_raw"])
        assert float((quat_wxyz_to_rotmat(m.get_rotation) - want).abs().max()) <= 2e-6
        assert float((m.get_rotation.norm(dim=1) - 1).abs().max()) < 1e-6
        assert close(m.get_xyz, d["get_xyz"], 2e-6, 2e-7) and close(m.get_scaling, d["get_scaling"], 2e-6, 1e-9)
        assert close(m.face_center, d["face_center"], 2e-6, 2e-7) and close(m.face_scaling, d["face_scaling"], 2e-6, 1e-9)
        assert close(m.get_opacity, d["get_opacity"], 2e-6, 1e-8) and close(m.get_features, d["get_features"], 1e-5, 1e-7)
        assert close(m.get_covariance(1.5), d["get_covariance_1p5"], 1e-5, 1e-8)      # (off-diagonal entries cancel: absolute floor)
    lrs = [m.update_learning_rate(int(i)) for i in d["lr_iters"]]
    assert np.allclose(np.array(lrs), d["lr_values"], rtol=1e-12, atol=0)
    assert np.allclose(np.array([g["lr"] for g in m.optimizer.param_groups]), d["lr_groups_after"], rtol=1e-12, atol=0)
    before = m._opacity
    m.reset_opacity()
    assert m._opacity is before                                    # in place here (the reference swaps in a new Parameter)
    assert close(m._opacity, d["reset_opacity"], 1e-5, 1e-6)
    st = m.optimizer.state[m._opacity]
    assert not bool(st["exp_avg"].any()) and not bool(st["exp_avg_sq"].any())
    assert not d["reset_opacity_m1"].any() and not d["reset_opacity_m2"].any()
    assert float(m.optimizer.state[m._scaling]["exp_avg"].abs().max()) > 0 and float(d["reset_other_m1_absmax"]) > 0
