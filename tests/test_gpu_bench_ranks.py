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
