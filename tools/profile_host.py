"""Host-side cost of the two EAGER loop shapes of bench.py (GPU box): cProfile (tottime) of
  loop : update_face_coor -> render() -> (image * w).sum().backward()          (bench: per_view_loop_views_per_sec)
  step : ggsplat.inner_step.registration_step(fused_loss=True) with torch Adam  (bench: s2_inner_step_iters_per_sec)
plus the wall-clock rate of each (synchronised once per pass).  Usage: python tools/profile_host.py [loop|step] [n_rows]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace  # noqa: E402

from ggsplat import synthetic as S  # noqa: E402
from ggsplat.inner_step import DEFAULT_OPT, registration_step  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel  # noqa: E402
from ggsplat.render import render  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "step"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda"
W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)[:32]
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); w = torch.randn(3, H, W, device=dev)
with torch.no_grad():
    m.update_face_coor()
    gts = [(render(c, m, pipe, bg)["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp_(0, 1).contiguous() for c in cams]
mask = (torch.rand(1, H, W, device=dev) > 0.1).float()
m.training_setup(DEFAULT_OPT, is_ff=True)


def loop():
    for c in cams:
        m.update_face_coor()
        pkg = render(c, m, pipe, bg)
        (pkg["render"] * w).sum().backward()
        for q in m.parameters():
            q.grad = None


def step():
    for c, gt in zip(cams, gts):
        registration_step(m, c, gt, mask, bg, fused_loss=True)


fn = loop if which == "loop" else step
for _ in range(2):
    fn()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{which}: {len(cams) / best:.0f} per second ({best / len(cams) * 1e6:.0f} us each), best of 3 passes over {len(cams)} cameras")
# host time only: how long the Python side takes to ISSUE one pass (no sync inside)
t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"{which}: host issue time {(t1 - t0) / len(cams) * 1e6:.0f} us per iteration (includes the rasterizer's header wait)")
pr = cProfile.Profile(); pr.enable(); fn(); fn(); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(rows)
