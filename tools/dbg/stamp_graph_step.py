"""The captured s2 iteration (config 2: 100k Gaussians, 1080p, one view) with the library's TIMESTAMP profile captured inside the graph
(ggsplat.profile.DeviceStamps): per-kernel microseconds of the replayed iteration itself, no profiler attached, next to the iteration
rate.  `python tools/dbg/stamp_graph_step.py [N]`; GGS_LIB_PATH selects a variant library (what-if builds)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace
from ggsplat import synthetic as S
from ggsplat.adam import GraphAdam
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
from ggsplat.mesh_gaussian_model import MeshGaussianModel
from ggsplat.profile import DeviceStamps, NAMES
from ggsplat.render import render
dev, W, H = "cuda", 1920, 1080
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
m.training_setup(DEFAULT_OPT, is_ff=True)
m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
cams = S.rig_cameras(device=dev)[:16]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
with torch.no_grad():
    m.update_face_coor()
    gts = [(render(c, m, pipe, bg)["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp(0, 1).contiguous() for c in cams]
mask = (torch.rand(1, H, W, device=dev) > 0.1).float()


def rate(step, n):
    for i in range(2):
        step(cams[i], gts[i], mask)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        step(cams[i % 16], gts[i % 16], mask)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


plain = GraphedRegistrationStep(m, W, H, bg)
dt_plain = rate(plain, n)
print("un-instrumented: last iteration num_rendered", int(plain._hdr_host[0]), "overflow", int(plain._hdr_host[1]))
del plain
st = DeviceStamps(dev).start()
step = GraphedRegistrationStep(m, W, H, bg)
step(cams[0], gts[0], mask)            # warm-up + capture (the capture restarts the stamp numbering) + first replay
st.stop()
acc = {k: 0.0 for k in NAMES}
span = between = 0.0
t0 = time.perf_counter()
for i in range(n):
    out = step(cams[i % 16], gts[i % 16], mask)    # every call waits for its result: the slots are complete afterwards
    r = st.read()
    for k in NAMES:
        acc[k] += r["seconds"][k]
    span += r["span"]; between += r["between_brackets"]
    if i in (0, 1, 2, 7, 31, n - 1):
        print(f"  iteration {i}: num_rendered {int(step._hdr_host[0])} overflow {int(step._hdr_host[1])} loss {out['loss']:.5f} "
              f"span {r['span'] * 1e6:.1f} us scatter {r['seconds']['scatter'] * 1e6:.1f} us")
dt_inst = (time.perf_counter() - t0) / n
print(f"captured s2 iteration, library {os.path.basename(os.environ.get('GGS_LIB_PATH', 'libggsplat.so'))}: {dt_plain * 1e6:.1f} us per iteration "
      f"({1 / dt_plain:.0f} it/s); with {r['n_stamps']} stamps captured inside {dt_inst * 1e6:.1f} us (host reads the slots every iteration)")
print("in-graph us per iteration: " + ", ".join(f"{k} {acc[k] / n * 1e6:.1f}" for k in NAMES if acc[k] > 0)
      + f" | first to last stamp {span / n * 1e6:.1f}, between brackets (loss, regularisers, Adam, launch gaps) {between / n * 1e6:.1f}")
