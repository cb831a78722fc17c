"""PLY persistence in the reference's layout (CPU): header / property order, binary round trip, SH block
transposition, ascii reading."""
import numpy as np
import pytest
import torch

from ggsplat import ply_io


def _params(P=17, K=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(_xyz=torch.randn(P, 3, generator=g), _features_dc=torch.randn(P, 1, 3, generator=g),
                _features_rest=torch.randn(P, K - 1, 3, generator=g), _opacity=torch.randn(P, 1, generator=g),
                _scaling=torch.randn(P, 3, generator=g), _rotation=torch.randn(P, 4, generator=g))


def test_property_order_matches_reference():
    names = ply_io.attribute_names(3, 45)
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44"
    assert names[54:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(names) == 62                       # the 62-float record of a degree-3 3DGS point cloud


def test_binary_round_trip_and_header(tmp_path):
    p = _params()
    path = str(tmp_path / "pc" / "local_point_cloud.ply")
    ply_io.save_gaussians(path, **p)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode().splitlines()
    assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0" and head[2] == "element vertex 17"
    assert [l.split()[-1] for l in head[3:]] == ply_io.attribute_names(3, 45)
    assert all(l.startswith("property float ") for l in head[3:])
    assert len(raw) == raw.index(b"end_header\n") + 11 + 17 * 62 * 4
    back = ply_io.load_gaussians(path, max_sh_degree=3, device="cpu")
    for k, v in p.items():
        assert torch.equal(back[k], v), k
    # SH blocks are stored channel-major: f_rest_0..14 = red coefficients 1..15
    d = ply_io.read_ply(path)
    assert np.array_equal(d["f_rest_0"], p["_features_rest"][:, 0, 0].numpy())
    assert np.array_equal(d["f_rest_15"], p["_features_rest"][:, 0, 1].numpy())
    assert np.array_equal(d["nx"], np.zeros(17, np.float32))


def test_degree_mismatch_and_ascii(tmp_path):
    p = _params(K=4)
    path = str(tmp_path / "a.ply")
    ply_io.save_gaussians(path, **p)
    with pytest.raises(AssertionError):
        ply_io.load_gaussians(path, max_sh_degree=3, device="cpu")
    txt = tmp_path / "t.ply"
    txt.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty uchar red\n"
                   "end_header\n1.5 2.5 7\n-1 0 255\n")
    d = ply_io.read_ply(str(txt))
    assert d["x"].tolist() == [1.5, -1.0] and d["red"].tolist() == [7, 255] and d["red"].dtype == np.uint8


@pytest.mark.parametrize("tag", ["all", "valid"])
def test_save_ply_writes_what_the_reference_hands_to_plyfile(tag, tmp_path):
    """MeshGaussianModel.save_ply(path, save_local=True) (row f2) against a RECORDING of the reference's own save_ply
    (scene/mesh_gaussian_model.py:251-283; tests/golden/ply_layout.npz: the structured array it hands to plyfile and the binding.pkl
    it pickles): property order, every per-vertex record, the binding of the written Gaussians, with and without `valid_faces`."""
    import os
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    from ggsplat.mesh_io import load_binding
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "ply_layout.npz"))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    params = {k: torch.tensor(d["p" + k]) for k in names}
    params["binding"] = torch.tensor(d["binding"])
    P = params["binding"].shape[0]
    verts = torch.randn(60, 3, generator=torch.Generator().manual_seed(0))
    faces = torch.randint(0, 60, (40, 3), generator=torch.Generator().manual_seed(1))
    m = MeshGaussianModel.from_tensors(verts, faces, params, sh_degree=1, device="cpu")
    if tag == "valid":
        m.mesh.valid_faces = [int(x) for x in d["valid_faces"]]
    path = str(tmp_path / "pc" / "local_point_cloud.ply")
    m.save_ply(path, save_local=True)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode().splitlines()
    assert [l.split()[-1] for l in head[3:]] == [str(n) for n in d[f"{tag}_names"]]
    cols = ply_io.read_ply(path)
    got = np.stack([cols[str(n)] for n in d[f"{tag}_names"]], 1)
    assert got.shape == d[f"{tag}_records"].shape and np.array_equal(got, d[f"{tag}_records"])
    assert np.array_equal(load_binding(os.path.join(os.path.dirname(path), "binding.pkl"), device="cpu").numpy(), d[f"{tag}_binding_pkl"])
    assert got.shape[0] == (P if tag == "all" else int(np.isin(d["binding"], d["valid_faces"]).sum()))
