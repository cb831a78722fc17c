"""bench.py's multi-rank path, driven the way the driver drives it: `python bench.py --gpus N` with NO launcher around
it must start N ranks by itself (the reference has no distributed path:
s2_registration.py:241-251 is single-process).

One GPU is enough: both ranks use cuda:0 (--single-device) and exchange gradients over gloo; the code path is the
one RCCL runs (init_process_group, the captured step followed by an in-place all-reduce on the graph's static output,
the barrier / MAX reduce of the timing).  Checks: n_gpus == 2, both ranks listed with their view shards, and the
all-reduced gradient bucket equals the 1-rank bucket (the sum over views does not depend on the sharding beyond fp32
summation order).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--views", "8", "--cpu-views", "0", "--loop-views", "0", "--width", "640",
          "--height", "360", "--n-around", "80", "--n-rows", "60"]


def _run(extra, tmp_path, name):
    dump = str(tmp_path / f"{name}.pt")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra + ["--dump-grads", dump],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0]), torch.load(dump)


@pytest.mark.parametrize("graph", [True, False])
def test_bench_spawns_its_ranks_and_sums_gradients(tmp_path, graph):
    extra = [] if graph else ["--no-graph"]
    one, g1 = _run(["--gpus", "1"] + extra, tmp_path, "one")
    two, g2 = _run(["--gpus", "2", "--backend", "gloo", "--single-device"] + extra, tmp_path, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    ranks = two["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["views"] for r in ranks] == [4, 4]
    assert two["config"]["views_per_step"] == 8 and two["scaling"] == "strong"
    assert g1.shape == g2.shape and float(g1.abs().sum()) > 0
    rel = float((g1.double() - g2.double()).abs().sum() / g1.double().abs().sum())
    assert rel < 1e-6, rel


@pytest.mark.parametrize("graph", [True, False])
def test_overlapped_gradient_exchange_gives_the_same_bucket(tmp_path, graph):
    """`--overlap 2`: every rank renders its views in two slices and all-reduces slice 0 while slice 1 renders
    (ggsplat.dist.all_reduce_parts).  Same bucket as the single all-reduce and as one rank, up to fp32 summation order; the
    line says how the exchange was done.  Also with ONE rank (two captured slices, no collective)."""
    extra = [] if graph else ["--no-graph"]
    one, g1 = _run(["--gpus", "1"] + extra, tmp_path, "one")
    two, g2 = _run(["--gpus", "2", "--backend", "gloo", "--single-device", "--overlap", "2"] + extra, tmp_path, "two_overlap")
    solo, g3 = _run(["--gpus", "1", "--overlap", "2"] + extra, tmp_path, "one_overlap")
    assert two["n_gpus"] == 2 and two["config"]["all_reduce_parts"] == 2 and "overlapped" in two["config"]["collective"]
    assert one["config"]["all_reduce_parts"] == 1 and one["config"]["collective"].startswith("none")
    assert solo["config"]["all_reduce_parts"] == 2
    for g in (g2, g3):
        assert g.shape == g1.shape
        rel = float((g1.double() - g.double()).abs().sum() / g1.double().abs().sum())
        assert rel < 1e-6, rel


def test_two_rank_bucket_against_the_oracle_at_config3_size(tmp_path):
    """BASELINE configs[2] at its own size: 100 000 mesh-bound Gaussians, 1920x1080, the inner loop of
    s2_registration.py:238-327 with the views sharded over TWO ranks (views[rank::2], one captured step per rank, one
    all-reduce of the flat bucket [mesh.v | _xyz | f_dc | f_rest | opacity | scaling | rotation]).  The all-reduced bucket
    rank 0 dumps is compared, tensor by tensor, with the ORACLE pipeline evaluated independently on the host:
    host_oracle.mesh_bind (autograd) -> C oracle forward + backward of every view with the same seeded dL/dimage ->
    gradients summed over the views in fp64 -> autograd back to the parameters and mesh.v.  <= 1e-4 relative L1 each."""
    import math

    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
    from ggsplat import synthetic as S
    from oracle import host_oracle as HO
    from oracle.c_oracle import COracle

    n_views, W, H = 8, 1920, 1080
    dump = str(tmp_path / "two_rank.pt")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--views", str(n_views), "--steps", "1", "--warmup", "1", "--cpu-views", "0", "--loop-views", "0",
                        "--extra-configs", "0", "--dump-grads", dump], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and [r["views"] for r in line["config"]["ranks"]] == [4, 4]
    assert "100000 mesh-bound" in line["config"]["workload"] and "1920x1080" in line["config"]["workload"]
    flat = torch.load(dump).double()

    # the same scene, built the way bench.py builds it
    verts, faces = S.skirt_mesh(200, 250)
    Fn = faces.shape[0]
    params = S.skirt_gaussian_params(Fn, sh_degree=0)
    cams = S.rig_cameras(n_rings=1, n_az=n_views, width=W, height=H, f=1500.0)[:n_views]
    w_img = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1234))
    names = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]
    leaf = {n: params[n].clone().requires_grad_(True) for n in names}
    mv = verts.clone().requires_grad_(True)
    xyz, scaling, rot = HO.mesh_bind(mv, faces, params["binding"], leaf["_xyz"], leaf["_scaling"], leaf["_rotation"])
    opacity = torch.sigmoid(leaf["_opacity"])
    shs = torch.cat((leaf["_features_dc"], leaf["_features_rest"]), 1)
    keys = ("means3D", "scales", "rotations", "opacities", "shs")
    tensors = (xyz, scaling, rot, opacity, shs)
    acc = {k: np.zeros(tuple(t.shape), np.float64) for k, t in zip(keys, tensors)}
    for c in cams:
        co = COracle(means3D=xyz.detach(), opacities=opacity.detach(), shs=shs.detach(), scales=scaling.detach(),
                     rotations=rot.detach(), viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform,
                     campos=c.camera_center, bg=torch.zeros(3), W=W, H=H, tanfovx=math.tan(c.FoVx * 0.5),
                     tanfovy=math.tan(c.FoVy * 0.5), sh_degree=0)
        g = co.backward(w_img)
        for k in keys:
            acc[k] += np.asarray(g[k], np.float64).reshape(acc[k].shape)
        co.close()
    torch.autograd.backward(list(tensors), [torch.from_numpy(acc[k]).float() for k in keys])
    ref = [mv.grad] + [leaf[n].grad if leaf[n].grad is not None else torch.zeros_like(leaf[n]) for n in names]
    assert sum(r.numel() for r in ref) == flat.numel()
    o = 0
    errs = {}
    for name, r in zip(["mesh.v"] + names, ref):
        n = r.numel()
        if n:
            a = flat[o:o + n]
            errs[name] = float((a - r.reshape(-1).double()).abs().sum() / (r.double().abs().sum() + 1e-30))
        o += n
    print("\n[config 3, 2 ranks x 4 views, 100k / 1080p] all-reduced bucket vs oracle pipeline, relative L1:",
          {k: f"{v:.2e}" for k, v in errs.items()})
    assert float(flat.abs().sum()) > 0
    for name, e in errs.items():
        assert e <= 1e-4, (name, e)


def _full_size(extra, tmp_path, name, timeout=1500):
    dump = str(tmp_path / f"{name}.pt")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-views", "0", "--loop-views", "0",
                        "--extra-configs", "0", "--dump-grads", dump] + extra, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), torch.load(dump)


def test_eight_and_seven_ranks_at_the_real_shape_on_one_device(tmp_path):
    """The shape the driver's 8-GPU run has, rehearsed on ONE device over gloo: BASELINE configs[2] in full -- 100 000
    Gaussians, 160 views of 1920x1080 --, `python bench.py --gpus 8` spawning its own ranks, 20 views per rank in ONE launch set, the
    compute part captured into a hipGraph, one all-reduce of the flat bucket, the overflow flag and the capture-failed flag reduced
    over the ranks.  Then SEVEN ranks: shards of 23 and 22 views (views[rank::7]), so the flag / timing reductions and the launch
    sets see uneven shards.  The reduced bucket must equal the one-process 160-view bucket (fp32 summation order only), and the line
    must carry the contract's keys with every rank listed."""
    one, g1 = _full_size(["--gpus", "1"], tmp_path, "one")
    assert one["n_gpus"] == 1 and one["config"]["views_per_step"] == 160 and float(g1.abs().sum()) > 0
    for world, shards in ((8, [20] * 8), (7, [23, 23, 23, 23, 23, 23, 22])):
        many, g = _full_size(["--gpus", str(world), "--backend", "gloo", "--single-device"], tmp_path, f"w{world}")
        assert many["n_gpus"] == world and many["scaling"] == "strong" and many["config"]["views_per_step"] == 160
        assert [r["rank"] for r in many["config"]["ranks"]] == list(range(world))
        assert [r["views"] for r in many["config"]["ranks"]] == shards
        assert many["config"]["views_per_launch"] == min(40, shards[0]) and many["config"]["collective"] == "one all-reduce per step"
        assert many["value"] > 0 and many["roofline"]["kernel"] == "ggs_k_render_bwd" and many["cpu_baseline"] is None
        assert many["roofline"]["in_graph"] is not None and many["roofline"]["launches_per_step"] == 1
        assert g.shape == g1.shape
        rel = float((g1.double() - g.double()).abs().sum() / g1.double().abs().sum())
        assert rel < 2e-6, (world, rel)


def test_two_ranks_with_pipelined_launch_sets(tmp_path):
    """Two ranks x 80 views = two launch sets per rank, software-pipelined over a second stream inside each rank's captured
    graph (the default of every world size since round 6), against the serial launch sets and against one rank."""
    one, g1 = _full_size(["--gpus", "1", "--pipeline", "0"], tmp_path, "one_serial")
    two, g2 = _full_size(["--gpus", "2", "--backend", "gloo", "--single-device"], tmp_path, "two_default")
    assert two["config"]["launch_set_pipeline"] == 1 and [r["views"] for r in two["config"]["ranks"]] == [80, 80]
    rel = float((g1.double() - g2.double()).abs().sum() / g1.double().abs().sum())
    assert rel < 2e-6, rel
