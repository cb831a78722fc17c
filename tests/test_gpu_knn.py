"""distCUDA2 replacement (ggs_dist2_3nn) vs an exact KD-tree on the host: mean squared distance to the
3 nearest neighbours, self excluded, duplicates at distance 0 counted."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu


def _ref(pts):
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(1)


@pytest.mark.parametrize("P", [4, 257, 5000, 100000])
def test_dist2_matches_kdtree(P):
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(P)
    pts = torch.randn(P, 3, generator=g)
    out = distCUDA2(pts.cuda()).cpu().numpy()
    ref = _ref(pts.numpy())
    assert np.allclose(out, ref, rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("kind", ["gauss", "surface", "clustered", "planar", "line", "identical"])
def test_grid_search_equals_brute_force_bit_for_bit(kind):
    """ggs_dist2_3nn_grid against the O(P^2) kernel: the same fp32 distance expression over a superset of the true
    neighbours -> identical bits, whatever the cloud looks like (volume, thin surface like the garment, clusters with far
    outliers, degenerate extents, all points equal) -- and against the KD-tree."""
    from simple_knn._C import distCUDA2
    from ggsplat import synthetic as S
    g = torch.Generator().manual_seed(7)
    if kind == "gauss":
        pts = torch.randn(60_000, 3, generator=g)
    elif kind == "surface":
        v, f = S.skirt_mesh()
        pts = v[f].mean(1)                                                   # 100k face centres of the skirt tube
    elif kind == "clustered":
        pts = torch.cat([torch.randn(20_000, 3, generator=g) * 0.01, torch.randn(20_000, 3, generator=g) * 0.01 + 5.0,
                         torch.randn(50, 3, generator=g) * 100.0])
    elif kind == "planar":
        pts = torch.cat([torch.rand(30_000, 2, generator=g), torch.zeros(30_000, 1)], 1)
    elif kind == "line":
        pts = torch.cat([torch.rand(5_000, 1, generator=g), torch.full((5_000, 2), 0.25)], 1)
    else:
        pts = torch.full((3_000, 3), 1.5)
    a = distCUDA2(pts.cuda())
    b = distCUDA2(pts.cuda(), brute_force=True)
    assert torch.equal(a, b)
    if kind != "identical":
        assert np.allclose(a.cpu().numpy(), _ref(pts.numpy()), rtol=2e-5, atol=1e-9)


def test_duplicates_and_model_init():
    from simple_knn._C import distCUDA2
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    pts = torch.tensor([[0.0, 0, 0], [0.0, 0, 0], [1.0, 0, 0], [0.0, 2, 0], [5.0, 5, 5]])
    out = distCUDA2(pts.cuda()).cpu()
    assert abs(float(out[0]) - (0 + 1 + 4) / 3) < 1e-6          # its duplicate is a neighbour at distance 0
    m = MeshGaussianModel(0)
    sc = m.init_scaling_from_neighbours(pts.cuda())
    assert sc.shape == (5, 3) and torch.allclose(sc[:, 0], torch.log(torch.sqrt(out.clamp_min(1e-7))).cuda())


def test_ply_round_trip_through_model(tmp_path):
    from ggsplat import synthetic as S
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f = S.skirt_mesh(12, 6)
    p = S.skirt_gaussian_params(f.shape[0], sh_degree=2)
    m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=2, device="cuda")
    path = str(tmp_path / "point_cloud" / "frame_00000" / "local_point_cloud.ply")
    m.save_ply(path, save_local=True)
    m2 = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], sh_degree=2, seed=9), sh_degree=2, device="cuda")
    m2.load_ply(path)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(m2, k).cpu(), getattr(m, k).detach().cpu()), k
    m.update_face_coor(); m2.update_face_coor()
    assert torch.equal(m.get_xyz, m2.get_xyz) and torch.equal(m.get_rotation, m2.get_rotation)


def test_world_frame_ply_holds_the_getters(tmp_path):
    """save_ply(save_local=False) (scene/mesh_gaussian_model.py:261-264): world positions, log of the world scaling,
    world rotation -- i.e. what the fused mesh-binding kernel produces."""
    from ggsplat import ply_io, synthetic as S
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    v, f = S.skirt_mesh(12, 6)
    p = S.skirt_gaussian_params(f.shape[0], sh_degree=1)
    m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=1, device="cuda")
    m.mesh.valid_faces = list(range(0, f.shape[0], 2))            # only Gaussians on the valid faces are written
    path = str(tmp_path / "point_cloud" / "frame_00003" / "point_cloud.ply")
    m.save_ply(path)
    d = ply_io.load_gaussians(path, 1, device="cpu")
    keep = torch.arange(0, f.shape[0], 2)
    m.update_face_coor()
    assert torch.equal(d["_xyz"], m.get_xyz.detach().cpu()[keep])
    assert torch.equal(d["_scaling"], torch.log(m.get_scaling).detach().cpu()[keep])
    assert torch.equal(d["_rotation"], m.get_rotation.detach().cpu()[keep])
    assert not (tmp_path / "point_cloud" / "frame_00003" / "binding.pkl").exists()


@pytest.mark.parametrize("P", [1, 3, 4, 5, 40])
def test_grid_entry_point_with_very_few_points(P):
    """ggs_dist2_3nn_grid called directly (the shim sends fewer than 64 points to the brute force): same bits as the brute
    force, missing neighbours counted as FLT_MAX like upstream."""
    import ctypes as C
    from ggsplat._lib import check, lib, ptr
    pts = torch.randn(P, 3, generator=torch.Generator().manual_seed(P)).cuda()
    L = lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a, b = torch.empty(P, device="cuda"), torch.empty(P, device="cuda")
    scratch = torch.empty(L.ggs_dist2_3nn_scratch_bytes(P), device="cuda", dtype=torch.uint8)
    check(L.ggs_dist2_3nn_grid(P, ptr(pts), ptr(a), ptr(scratch), stream), "grid")
    check(L.ggs_dist2_3nn(P, ptr(pts), ptr(b), stream), "brute")
    assert torch.equal(a, b)
    if P < 4:
        assert float(a.min()) > 1e37
