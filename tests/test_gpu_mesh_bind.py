"""GPU parity of the fused mesh-binding kernels (ggs_mesh_bind_forward / _backward) against the
PyTorch restatement of the reference's getters (oracle/host_oracle.py; autograd for the backward)."""
import pytest
import torch

from helpers import REL_L1_TOL, rel_l1
from ggsplat import synthetic as S
from oracle import host_oracle as HO

pytestmark = pytest.mark.gpu


def _case(n_around, n_rows, seed, with_bary, multi_bind):
    v, f = S.skirt_mesh(n_around, n_rows, seed=seed)
    Fn = f.shape[0]
    g = torch.Generator().manual_seed(seed)
    if multi_bind:
        binding = torch.randint(0, Fn, (Fn * 2,), generator=g)
    else:
        binding = torch.arange(Fn)
    P = binding.shape[0]
    local = torch.randn(P, 3, generator=g) * 0.3
    ls = torch.randn(P, 3, generator=g) * 0.3 - 0.5
    rr = torch.randn(P, 4, generator=g)
    bary = None
    if with_bary:
        b = torch.rand(P, 3, generator=g)
        bary = b / b.sum(1, keepdim=True)
    return v, f, binding, local, ls, rr, bary


@pytest.mark.parametrize("with_bary,multi_bind", [(False, False), (True, True)])
def test_mesh_bind_forward_backward(with_bary, multi_bind):
    from ggsplat.mesh_gaussian_model import mesh_bind
    v, f, binding, local, ls, rr, bary = _case(24, 17, 3, with_bary, multi_bind)
    P = binding.shape[0]
    g = torch.Generator().manual_seed(5)
    wx, wsc, wr = torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)

    def run(fn, dev):
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (v, local, ls, rr)]
        xyz, sc, rot = fn(leaves[0], f.to(dev), binding.to(dev), leaves[1], leaves[2], leaves[3],
                          None if bary is None else bary.to(dev))
        ((xyz * wx.to(dev)).sum() + (sc * wsc.to(dev)).sum() + (rot * wr.to(dev)).sum()).backward()
        return [t.detach().cpu() for t in (xyz, sc, rot)], [t.grad.detach().cpu() for t in leaves]

    out_h, grad_h = run(mesh_bind, "cuda")
    out_o, grad_o = run(HO.mesh_bind, "cpu")
    for a, b, name in zip(out_h, out_o, ("xyz", "scaling", "rotation")):
        assert rel_l1(a, b) <= REL_L1_TOL, name
    for a, b, name in zip(grad_h, grad_o, ("dverts", "dlocal", "dlog_scaling", "draw_rot")):
        assert rel_l1(a, b) <= REL_L1_TOL, name


def test_all_quaternion_branches_hit():
    """The rotmat->quat construction has 4 branches; a closed tube exercises all of them."""
    v, f = S.skirt_mesh(64, 8, seed=0)
    R, _ = HO.compute_face_orientation(v, f)
    dec = torch.cat([R.diagonal(dim1=1, dim2=2), R.diagonal(dim1=1, dim2=2).sum(1, keepdim=True)], 1)
    assert set(dec.argmax(1).tolist()) == {0, 1, 2, 3}


def test_model_getters_and_face_quantities():
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f = S.skirt_mesh(20, 10, seed=1)
    params = S.skirt_gaussian_params(f.shape[0], sh_degree=1, seed=1)
    m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=1, device="cuda")
    m.update_face_coor()
    xyz, sc, rot = HO.mesh_bind(v, f, params["binding"], params["_xyz"], params["_scaling"], params["_rotation"])
    assert rel_l1(m.get_xyz, xyz) <= REL_L1_TOL and rel_l1(m.get_scaling, sc) <= REL_L1_TOL
    assert rel_l1(m.get_rotation, rot) <= REL_L1_TOL
    assert m.get_features.shape == (f.shape[0], 4, 3) and m.get_opacity.shape == (f.shape[0], 1)
    R, s = HO.compute_face_orientation(v, f)
    assert rel_l1(m.face_center, v[f].mean(1)) <= REL_L1_TOL
    assert rel_l1(m.face_scaling, s) <= REL_L1_TOL
