"""Stage-2 output directory round trip on the CPU: OBJ mesh IO in the dialect of
utils/io_utils.py:7-60, binding.pkl (scene/mesh_gaussian_model.py:277-282, :335-337) and the local PLY, laid out as
scene/scene.py:183-192 lays them out."""
import os
import pickle

import numpy as np
import pytest
import torch

from ggsplat import mesh_io, synthetic as S
from ggsplat.mesh_gaussian_model import MeshGaussianModel


def test_obj_round_trip_plain_and_textured(tmp_path):
    g = np.random.default_rng(0)
    verts = g.standard_normal((7, 3)).astype(np.float32)
    faces = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]])
    path = str(tmp_path / "meshes" / "frame_00000.obj")
    mesh_io.write_obj({"vertices": verts, "faces": faces}, path)
    text = open(path).read().splitlines()
    assert text[0].startswith("v ") and text[7] == "f 1 2 3" and text[-1] == "f 5 6 7"      # 1-based, no slashes
    d = mesh_io.read_obj(path)
    assert d["vertices"].dtype == np.float32 and np.array_equal(d["vertices"], verts)       # bit-exact float32
    assert np.array_equal(d["faces"], faces) and d["uvs"].size == 0 and d["texture_faces"].size == 0
    uvs = g.random((9, 2)).astype(np.float32)
    tfaces = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]])
    mesh_io.write_obj({"vertices": verts, "uvs": uvs, "faces": faces, "texture_faces": tfaces}, path)
    assert "f 1/1 2/2 3/3" in open(path).read()
    d = mesh_io.read_obj(path)
    assert np.array_equal(d["uvs"], uvs) and np.array_equal(d["texture_faces"], tfaces) and np.array_equal(d["faces"], faces)


def test_obj_reader_accepts_foreign_files(tmp_path):
    path = str(tmp_path / "t.obj")
    open(path, "w").write("# comment\n\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.25\nvt 1 0\nvt 0 1\n"
                          "g grp\nf 1/1/1 2/2/1 3/3/1\n")
    d = mesh_io.read_obj(path)
    assert d["vertices"].shape == (3, 3) and np.array_equal(d["faces"], [[0, 1, 2]])
    assert np.array_equal(d["texture_faces"], [[0, 1, 2]]) and d["uvs"].shape == (3, 2)


def test_stage2_frame_round_trip(tmp_path):
    v, f = S.skirt_mesh(12, 6)
    p = S.skirt_gaussian_params(f.shape[0], sh_degree=1)
    p["binding"] = torch.randperm(f.shape[0], generator=torch.Generator().manual_seed(0))     # not the identity
    m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=1, device="cpu")
    seq = str(tmp_path / "stage2" / "Take1")
    paths = mesh_io.frame_paths(seq, 12)
    assert paths["local_ply"].endswith(os.path.join("point_cloud", "frame_00012", "local_point_cloud.ply"))
    assert paths["mesh"].endswith(os.path.join("meshes", "frame_00012.obj"))
    m.save_ply(paths["local_ply"], save_local=True)
    m.save_mesh(paths["mesh"])
    assert os.path.exists(paths["binding"])
    raw = pickle.load(open(paths["binding"], "rb"))          # what the reference's pickle.load would see
    assert torch.is_tensor(raw) and raw.dtype == torch.int64 and torch.equal(raw, p["binding"])

    m2 = MeshGaussianModel(sh_degree=1)
    m2.load_mesh(paths["mesh"], device="cpu")
    m2.load_ply(paths["local_ply"])
    assert torch.equal(m2.mesh.v.detach(), m.mesh.v.detach()) and torch.equal(m2.mesh.f, m.mesh.f)
    assert m2.mesh.v.requires_grad and not m2._xyz.requires_grad            # load_ply: mesh trainable, Gaussians fixed
    assert torch.equal(m2.binding, p["binding"]) and m2.active_sh_degree == 1
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(m2, k), getattr(m, k).detach()), k
    assert m2.max_radii2D.shape == (f.shape[0],)


def test_valid_faces_filter_and_missing_binding(tmp_path):
    v, f = S.skirt_mesh(8, 4)
    p = S.skirt_gaussian_params(f.shape[0], sh_degree=0)
    m = MeshGaussianModel.from_tensors(v, f, p, sh_degree=0, device="cpu")
    m.mesh.valid_faces = [1, 5, 6]
    path = str(tmp_path / "pc" / "local_point_cloud.ply")
    m.save_ply(path, save_local=True)
    b = mesh_io.load_binding(os.path.join(os.path.dirname(path), "binding.pkl"), device="cpu")
    assert b.tolist() == [1, 5, 6]
    os.remove(os.path.join(os.path.dirname(path), "binding.pkl"))
    m2 = MeshGaussianModel.from_tensors(v, f, p, sh_degree=0, device="cpu")
    with pytest.raises(FileNotFoundError):
        m2.load_ply(path)                       # 3 Gaussians in the file, a 64-entry binding in the model


def test_binding_pickle_of_other_writers(tmp_path):
    path = str(tmp_path / "binding.pkl")
    pickle.dump(np.array([3, 1, 2], dtype=np.int32), open(path, "wb"))
    assert mesh_io.load_binding(path, device="cpu").tolist() == [3, 1, 2]
    pickle.dump([0, 2], open(path, "wb"))
    assert mesh_io.load_binding(path, device="cpu").dtype == torch.int64


@pytest.mark.parametrize("tag", ["uv", "plain"])
def test_obj_dialect_matches_the_reference(tag, tmp_path):
    """mesh_io.write_obj / read_obj against the reference's own utils/io_utils.py:7-60 (tests/golden/obj_io.npz: the text its
    write_obj produced and what its read_obj parsed): this repo writes a file the reference parses to the same arrays and parses
    the reference's file to the arrays the reference parses."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "obj_io.npz"))
    mesh = {k: d["in_" + k] for k in (("vertices", "uvs", "faces", "texture_faces") if tag == "uv" else ("vertices", "faces"))}
    ref_path = str(tmp_path / "ref.obj")
    open(ref_path, "wb").write(d[tag + "_text"].tobytes())
    got = mesh_io.read_obj(ref_path)                                  # the reference's file, this repo's reader
    for k in ("vertices", "uvs", "faces", "texture_faces"):
        ref = d[f"{tag}_read_{k}"]
        assert got[k].shape == ref.shape or (got[k].size == 0 and ref.size == 0), k
        if ref.size:
            assert np.array_equal(got[k], ref), k
    ours = str(tmp_path / "ours.obj")
    mesh_io.write_obj(mesh, ours)                                     # this repo's file: parses to the same arrays
    back = mesh_io.read_obj(ours)
    for k in ("vertices", "uvs", "faces", "texture_faces"):
        ref = d[f"{tag}_read_{k}"]
        if ref.size:
            assert np.array_equal(back[k], ref), k
    # same statements, same order: v lines, vt lines, f lines with 1-based a/t pairs (float formatting may differ in digits)
    kinds = lambda p: [l.split()[0] for l in open(p).read().splitlines() if l.strip()]
    assert kinds(ours) == kinds(ref_path)
    f_ours = [l for l in open(ours).read().splitlines() if l.startswith("f ")]
    f_ref = [l for l in open(ref_path).read().splitlines() if l.startswith("f ")]
    assert f_ours == f_ref
