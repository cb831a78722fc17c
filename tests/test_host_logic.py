"""Host-side logic that needs no GPU: synthetic workloads, the render() mirror's selection logic
(with the native rasterizer stubbed out), camera helpers."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ggsplat import synthetic as S
from ggsplat import cameras as CAM


def test_config2_sizes():
    v, f = S.skirt_mesh()
    assert v.shape == (50200, 3) and f.shape == (100000, 3)
    assert int(f.max()) == 50199 and int(f.min()) == 0
    e1, e2 = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    assert float(torch.linalg.cross(e1, e2).norm(dim=1).min()) > 0   # no degenerate faces
    p = S.skirt_gaussian_params(100000, sh_degree=3)
    assert p["_features_rest"].shape == (100000, 15, 3) and p["binding"].dtype == torch.int64
    cams = S.rig_cameras()
    assert len(cams) == 160 and cams[0].image_width == 1920 and cams[0].image_height == 1080
    c = S.stack_cameras(cams[:3])
    assert c["view"].shape == (3, 16) and c["tanfov"].shape == (3, 2)
    assert abs(float(c["tanfov"][0, 0]) - 960 / 1500) < 1e-6


def test_look_at_camera_sees_target_at_principal_point():
    cam = CAM.look_at_camera((2.5, 0.7, 0.3), (0, 1, 0), width=1920, height=1080, fx=1500., fy=1500., cx=955., cy=548., device="cpu")
    hom = torch.tensor([[0.0, 1.0, 0.0, 1.0]]) @ cam.full_proj_transform
    ndc = hom[0, :2] / hom[0, 3]
    assert abs(float(((ndc[0] + 1) * 1920 - 1) / 2) - (955 - 0.5)) < 1e-2
    assert abs(float(((ndc[1] + 1) * 1080 - 1) / 2) - (548 - 0.5)) < 1e-2
    assert np.allclose(cam.camera_center.numpy(), [2.5, 0.7, 0.3], atol=1e-5)
    # +y of the image points down in the world
    up = torch.tensor([[0.0, 1.2, 0.0, 1.0]]) @ cam.full_proj_transform
    assert float(up[0, 1] / up[0, 3]) < float(ndc[1])


class _StubRasterizer:
    """Stands in for rasterize_gaussians (what GaussianRasterizer(settings)(...) calls): records settings + arguments."""
    calls = []

    @staticmethod
    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings):
        kw = dict(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                  scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        _StubRasterizer.calls.append((settings, kw))
        P = means3D.shape[0]
        H, W = settings.image_height, settings.image_width
        return torch.zeros(3, H, W), torch.arange(P, dtype=torch.int32) % 2, torch.zeros(1, H, W), torch.zeros(1, H, W)


@pytest.fixture
def render_mod(monkeypatch):
    from ggsplat import render as RM
    monkeypatch.setattr(RM, "rasterize_gaussians", _StubRasterizer.rasterize)
    _StubRasterizer.calls.clear()
    return RM


def _pc(P=6, K=4, with_shs=False, with_local=False):
    pc = SimpleNamespace(_xyz=torch.zeros(P, 3), active_sh_degree=1, max_sh_degree=1,
                         get_xyz=torch.randn(P, 3), get_opacity=torch.rand(P, 1), get_scaling=torch.rand(P, 3),
                         get_rotation=torch.randn(P, 4), get_features=torch.randn(P, K, 3),
                         get_covariance=lambda m: torch.full((P, 6), float(m)))
    if with_shs:
        pc.shs = torch.randn(P, K, 3)
    if with_local:
        pc.local_xyz = torch.randn(P, 3)
        pc.get_final_xyz = torch.randn(P, 3) + 10
    return pc


def test_render_mirror_default_path(render_mod):
    cam = S.orbit_cameras(1, width=64, img_height=48)[0]
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    pc = _pc()
    out = render_mod.render(cam, pc, pipe, torch.zeros(3))
    rs, kw = _StubRasterizer.calls[-1]
    # the reference's keys (gaussian_renderer/__init__.py:114-122) + "tile_count": the list lengths of THIS forward for the
    # region-of-interest loss (an extra key a caller of the reference's dict never looks at)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "3dposition", "depth", "alpha", "tile_count"}
    assert rs.image_height == 48 and rs.image_width == 64 and rs.sh_degree == 1 and rs.prefiltered is False
    assert abs(rs.tanfovx - math.tan(cam.FoVx * 0.5)) < 1e-12
    assert kw["shs"] is pc.get_features and kw["colors_precomp"] is None
    assert kw["scales"] is pc.get_scaling and kw["rotations"] is pc.get_rotation and kw["cov3D_precomp"] is None
    assert kw["means3D"] is pc.get_xyz and kw["means2D"].requires_grad
    assert out["visibility_filter"].dtype == torch.bool and out["viewspace_points"] is kw["means2D"]


def test_render_mirror_s3_selection_and_vis_mask(render_mod):
    cam = S.orbit_cameras(1, width=32, img_height=32)[0]
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    pc = _pc(with_shs=True, with_local=True)
    mask = torch.tensor([1, 0, 1, 1, 0, 0], dtype=torch.bool)
    out = render_mod.render(cam, pc, pipe, torch.zeros(3), vis_mask=mask)
    _, kw = _StubRasterizer.calls[-1]
    assert torch.equal(kw["means3D"], pc.get_final_xyz[mask])           # get_final_xyz when local_xyz is set
    assert torch.equal(kw["shs"], pc.shs[mask])                          # pc.shs wins over get_features
    assert kw["opacities"].shape == (3, 1) and kw["scales"].shape == (3, 3) and kw["means2D"].shape == (3, 3)
    assert out["radii"].shape == (3,)


def test_render_mirror_python_paths_and_override(render_mod):
    cam = S.orbit_cameras(1, width=32, img_height=32)[0]
    pc = _pc()
    pipe = SimpleNamespace(debug=True, compute_cov3D_python=True, convert_SHs_python=True)
    render_mod.render(cam, pc, pipe, torch.zeros(3), scaling_modifier=0.5)
    rs, kw = _StubRasterizer.calls[-1]
    assert rs.debug is True and rs.scale_modifier == 0.5
    assert kw["scales"] is None and torch.equal(kw["cov3D_precomp"], torch.full((6, 6), 0.5))
    assert kw["shs"] is None and kw["colors_precomp"].shape == (6, 3) and float(kw["colors_precomp"].min()) >= 0
    col = torch.rand(6, 3)
    render_mod.render(cam, pc, SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False),
                      torch.zeros(3), override_color=col)
    assert _StubRasterizer.calls[-1][1]["colors_precomp"] is col


def test_dropin_module_signature_and_errors():
    import diff_gaussian_rasterization_depth_alpha as D
    assert D.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    rs = D.GaussianRasterizationSettings(8, 8, 1., 1., torch.zeros(3), 1., torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    r = D.GaussianRasterizer(raster_settings=rs)
    z = torch.zeros(2, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=torch.zeros(2, 1), shs=torch.zeros(2, 1, 3), colors_precomp=z, scales=z, rotations=torch.zeros(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(means3D=z, means2D=z, opacities=torch.zeros(2, 1), shs=torch.zeros(2, 1, 3), scales=z, rotations=torch.zeros(2, 4), cov3D_precomp=torch.zeros(2, 6))
    vis = r.markVisible(torch.tensor([[0.0, 0, 1.0], [0.0, 0, 0.1]]))
    assert vis.tolist() == [True, False]


def test_model_schedule_and_sh_degree_surface():
    """update_learning_rate / oneupSHdegree of the model mirror (scene/gaussian_model.py:121-123, 171-177): only the
    group named "xyz" follows the schedule."""
    from types import SimpleNamespace
    from ggsplat.inner_step import DEFAULT_OPT
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    m = MeshGaussianModel(sh_degree=2)
    m.optimizer = SimpleNamespace(param_groups=[{"name": "xyz", "lr": 1.0}, {"name": "f_dc", "lr": 0.5}])
    from ggsplat.schedule import get_expon_lr_func
    m.xyz_scheduler_args = get_expon_lr_func(DEFAULT_OPT.position_lr_init, DEFAULT_OPT.position_lr_final,
                                             lr_delay_mult=DEFAULT_OPT.position_lr_delay_mult,
                                             max_steps=DEFAULT_OPT.position_lr_max_steps)
    lr0 = m.update_learning_rate(0)
    assert lr0 == pytest.approx(DEFAULT_OPT.position_lr_init) and m.optimizer.param_groups[0]["lr"] == lr0
    assert m.update_learning_rate(DEFAULT_OPT.position_lr_max_steps) == pytest.approx(DEFAULT_OPT.position_lr_final)
    assert m.optimizer.param_groups[1]["lr"] == 0.5
    assert m.active_sh_degree == 0
    for _ in range(5):
        m.oneupSHdegree()
    assert m.active_sh_degree == m.max_sh_degree == 2


def test_graph_optimiser_and_step_refuse_what_they_cannot_run():
    """No CPU path in the product: GraphAdam needs GPU parameters; the graphed step needs a GraphAdam."""
    import torch
    from ggsplat.adam import GraphAdam
    from ggsplat.inner_step import GraphedRegistrationStep
    with pytest.raises(RuntimeError):
        GraphAdam([{"params": [torch.zeros(4, requires_grad=True)], "lr": 1e-3}])
    from types import SimpleNamespace
    fake = SimpleNamespace(optimizer=object(), _xyz=torch.zeros(1, 3))
    with pytest.raises(TypeError):
        GraphedRegistrationStep(fake, 16, 16, torch.zeros(3))


def test_reset_opacity_in_place():
    """scene/gaussian_model.py:212-215: opacity clamped to <= 0.01, moments of that group zeroed; same tensor object."""
    import torch
    from types import SimpleNamespace
    from ggsplat.mesh_gaussian_model import MeshGaussianModel
    m = MeshGaussianModel(sh_degree=0)
    m._xyz = torch.zeros(5, 3)
    m._opacity = torch.tensor([[-6.0], [-4.0], [0.0], [2.0], [5.0]], requires_grad=True)
    m.optimizer = torch.optim.Adam([{"params": [m._opacity], "lr": 0.05, "name": "opacity"}], lr=0.0, eps=1e-15)
    m._opacity.grad = torch.ones_like(m._opacity)
    m.optimizer.step()
    ptr = m._opacity.data_ptr()
    before = torch.sigmoid(m._opacity.detach()).clone()
    m.reset_opacity()
    after = torch.sigmoid(m._opacity.detach())
    assert m._opacity.data_ptr() == ptr and m.num_gs == 5
    assert torch.allclose(after, torch.clamp(before, max=0.01), rtol=1e-5, atol=1e-7)
    st = m.optimizer.state[m._opacity]
    assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run with N ranks on
    127.0.0.1; under a launcher whose WORLD_SIZE disagrees with --gpus it refuses to print a line."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # launcher present but with another world size: no line, non-zero exit
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value.code)


def test_bench_failure_is_one_json_line_with_an_error_field():
    """A run that cannot be measured (here: no GPU in the CPU container, so the very first device call fails; on an 8-GPU node
    the same path catches an RCCL that cannot initialise) prints ONE parsable JSON line on rank 0 -- leading keys of the contract,
    value null, the reason and the stage in `error` -- and exits non-zero."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU to provoke the failure")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 1 and d["unit"] == "views/s" and d["metric"].startswith("fwd+bwd views/sec")
    assert d["error"].startswith("start-up: ") and "error_at" in d
    # a rank other than 0 prints nothing
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1"], capture_output=True,
                       text=True, timeout=300, env={**env, "WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1",
                                                     "MASTER_PORT": "29999"})
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_ctypes_mirrors_of_the_abi_structs_have_the_c_layout(tmp_path):
    """GgsParams / GgsStepPrologue / GgsStepTail are filled in Python and read in C: sizes and field offsets must agree
    (include/ggsplat.h compiled by gcc against the ctypes.Structure mirrors of ggsplat/_lib.py)."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    from ggsplat import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"GgsParams": _lib.GgsParams, "GgsStepPrologue": _lib.GgsStepPrologue, "GgsStepTail": _lib.GgsStepTail}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ggsplat.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fld, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fld} %zu\\n", offsetof({name}, {fld}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == C.sizeof(cls), name
        for fld, _ in cls._fields_:
            assert int(got[f"{name}.{fld}"]) == getattr(cls, fld).offset, f"{name}.{fld}"
