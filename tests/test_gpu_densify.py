"""Adaptive density control of the mesh-bound model (ggsplat.densify) against a plain-PyTorch restatement of the reference's
functions (scene/mesh_gaussian_model.py:130-208, scene/gaussian_model.py:276-408) on a 2k-Gaussian fixture, with
torch.optim.Adam and with GraphAdam, and the captured registration step re-capturing itself when P changes."""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from ggsplat import synthetic as S  # noqa: E402
from ggsplat.adam import GraphAdam  # noqa: E402
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel  # noqa: E402
from oracle import host_oracle as HO  # noqa: E402

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


class RefState:
    """The reference's algorithm on plain tensors (no optimiser object: moments are carried beside the parameters)."""

    def __init__(self, m):
        names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
        self.p = {n: getattr(m, n).detach().clone() for n in names}
        self.m1 = {n: m.optimizer.state[getattr(m, n)]["exp_avg"].clone() for n in names}
        self.m2 = {n: m.optimizer.state[getattr(m, n)]["exp_avg_sq"].clone() for n in names}
        self.verts, self.faces = m.mesh.v.detach().clone(), m.mesh.f.clone()
        self.binding = m.binding.clone()
        self.counter = torch.bincount(self.binding, minlength=self.faces.shape[0]).int()
        self.accum, self.denom, self.radii = m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone()
        self.percent_dense = m.percent_dense

    def bound(self):
        # the host oracle is a CPU restatement: bind there, compare on the GPU
        xyz, scaling, rot = HO.mesh_bind(self.verts.cpu(), self.faces.cpu(), self.binding.cpu(), self.p["_xyz"].cpu(),
                                         self.p["_scaling"].cpu(), self.p["_rotation"].cpu())
        return xyz.to(DEV), scaling.to(DEV), rot.to(DEV)

    def face_scaling(self):
        Fn = self.faces.shape[0]
        z = torch.zeros(Fn, 3)
        ident = torch.tensor([1.0, 0, 0, 0]).repeat(Fn, 1)
        return HO.mesh_bind(self.verts.cpu(), self.faces.cpu(), torch.arange(Fn), z, z, ident)[1][:, :1].to(DEV)

    def prune(self, mask):
        mask = mask.clone()
        b = self.binding[mask]
        cp = torch.zeros_like(self.counter)
        cp.scatter_add_(0, b, torch.ones_like(b, dtype=torch.int32))
        red = (self.counter - cp) > 0
        mask[mask.clone()] = red[b]
        keep = ~mask
        for d in (self.p, self.m1, self.m2):
            for n in d:
                d[n] = d[n][keep]
        self.accum, self.denom, self.radii = self.accum[keep], self.denom[keep], self.radii[keep]
        gone = self.binding[mask]
        self.counter.scatter_add_(0, gone, -torch.ones_like(gone, dtype=torch.int32))
        self.binding = self.binding[keep]

    def postfix(self, new, new_binding):
        self.binding = torch.cat((self.binding, new_binding))
        self.counter.scatter_add_(0, new_binding, torch.ones_like(new_binding, dtype=torch.int32))
        for n, t in new.items():
            self.p[n] = torch.cat((self.p[n], t))
            self.m1[n] = torch.cat((self.m1[n], torch.zeros_like(t)))
            self.m2[n] = torch.cat((self.m2[n], torch.zeros_like(t)))
        P = self.p["_xyz"].shape[0]
        self.accum, self.denom, self.radii = torch.zeros(P, 1, device=DEV), torch.zeros(P, 1, device=DEV), torch.zeros(P, device=DEV)

    def clone(self, grads, thr, extent):
        _, scaling, _ = self.bound()
        sel = (torch.norm(grads, dim=-1) >= thr) & (scaling.max(1).values <= self.percent_dense * extent)
        self.postfix({n: t[sel] for n, t in self.p.items()}, self.binding[sel])

    def split(self, grads, thr, extent, N=2):
        n0 = self.p["_xyz"].shape[0]
        pg = torch.zeros(n0, device=DEV)
        pg[:grads.shape[0]] = grads.squeeze()
        xyz, scaling, _ = self.bound()
        sel = (pg >= thr) & (scaling.max(1).values > self.percent_dense * extent)
        stds = scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=DEV), std=stds)
        r = self.p["_rotation"][sel]
        q = r / r.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                         1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                         1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3).repeat(N, 1, 1)
        new = {"_xyz": torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + xyz[sel].repeat(N, 1),
               "_scaling": torch.log((scaling[sel] / self.face_scaling()[self.binding[sel]]).repeat(N, 1) / (0.8 * N)),
               "_rotation": r.repeat(N, 1), "_features_dc": self.p["_features_dc"][sel].repeat(N, 1, 1),
               "_features_rest": self.p["_features_rest"][sel].repeat(N, 1, 1), "_opacity": self.p["_opacity"][sel].repeat(N, 1)}
        self.postfix(new, self.binding[sel].repeat(N))
        self.prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), device=DEV, dtype=torch.bool))))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        grads = self.accum / self.denom
        grads[grads.isnan()] = 0.0
        self.clone(grads, max_grad, extent)
        self.split(grads, max_grad, extent)
        _, scaling, _ = self.bound()
        mask = (torch.sigmoid(self.p["_opacity"]) < min_opacity).squeeze()
        if max_screen_size:
            mask = mask | (self.radii > max_screen_size) | (scaling.max(1).values > 0.1 * extent)
        self.prune(mask)


@pytest.mark.parametrize("graph_adam", [False, True])
@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_and_prune_matches_the_restated_reference(graph_adam, max_screen_size):
    m = small_model(graph_adam=graph_adam)
    ref = RefState(m)
    P0 = m._xyz.shape[0]
    extent = 3.0                                    # percent_dense * extent = 0.03 sits inside the fixture's range of max scales (0.02 .. 0.04)
    torch.manual_seed(123)
    ref.densify_and_prune(0.0002, 0.005, extent, max_screen_size)
    torch.manual_seed(123)
    m.densify_and_prune(0.0002, 0.005, extent, max_screen_size)
    P1 = m._xyz.shape[0]
    assert P1 == ref.p["_xyz"].shape[0] and P1 != P0
    for n, t in ref.p.items():
        got = getattr(m, n)
        assert isinstance(got, torch.nn.Parameter) and got.requires_grad and got.is_contiguous()
        tol = 2e-5 if n in ("_xyz", "_scaling") else 0.0        # children of a split go through the HIP binding (1-ulp noise)
        assert torch.allclose(got.detach(), t, rtol=tol, atol=tol), n
        st = m.optimizer.state[got]
        assert torch.equal(st["exp_avg"], ref.m1[n]) and torch.equal(st["exp_avg_sq"], ref.m2[n]), n
    assert torch.equal(m.binding, ref.binding)
    assert torch.equal(m.binding_counter, ref.counter)
    assert torch.equal(m.binding_counter, torch.bincount(m.binding, minlength=m.mesh.f.shape[0]).int())
    assert int(m.binding_counter.min()) >= 1                    # no face lost all its Gaussians
    assert m.xyz_gradient_accum.shape == (P1, 1) and m.denom.shape == (P1, 1) and m.max_radii2D.shape == (P1,)
    assert torch.equal(m.max_radii2D, ref.radii)
    # the optimiser drives the new tensors: a step changes them, and only them
    groups = {g["name"]: g["params"][0] for g in m.optimizer.param_groups}
    assert groups["xyz"] is m._xyz and groups["opacity"] is m._opacity and groups["vertex"] is m.mesh.v
    before = m._xyz.detach().clone()
    m._xyz.grad = torch.ones_like(m._xyz)
    m.optimizer.step()
    assert not torch.equal(before, m._xyz.detach())


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
