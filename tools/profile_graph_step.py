"""Graph-replayed s2 inner step at config 2 (100k Gaussians, 1080p, one view per iteration) -- run under
rocprofv3 --kernel-trace to see where the GPU time of one iteration goes.
  python tools/profile_graph_step.py [N] [--pipelined] [--silhouette [--plain-loss]]
--silhouette: per-camera garment masks (the initial model's alpha > 0.05, what a segmentation gives) instead of the 90 %-ones
salt-and-pepper mask; the captured step then picks the sparse-mask form of the loss's first pass (--plain-loss: keeps the plain one)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S
from ggsplat.adam import GraphAdam
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"
W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
m.training_setup(DEFAULT_OPT, is_ff=True)
m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
cams = S.rig_cameras(device=dev)[:16]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev)
# ground truth = the initial model's own render of each camera + a little noise: the optimisation stays near its starting
# point however many iterations are timed (against a random image the Gaussians grow without bound and the iteration with them)
from types import SimpleNamespace
from ggsplat.render import render
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
with torch.no_grad():
    m.update_face_coor()
    pkgs = [render(c, m, pipe, bg) for c in cams]
    gts = [(k["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp(0, 1).contiguous() for k in pkgs]
def flag(name):
    if name in sys.argv:
        sys.argv.remove(name)
        return True
    return False
pipelined, silhouette, plain = flag("--pipelined"), flag("--silhouette"), flag("--plain-loss")
if silhouette:
    masks = [(k["alpha"] > 0.05).float().reshape(1, H, W).contiguous() for k in pkgs]
else:
    masks = [(torch.rand(1, H, W, device=dev) > 0.1).float()] * len(cams)
del pkgs
kw = {"sparse_mask": False} if plain else {}
if pipelined:
    from ggsplat.inner_step import PipelinedRegistrationStep
    step = PipelinedRegistrationStep(m, W, H, bg, **kw)
else:
    step = GraphedRegistrationStep(m, W, H, bg, **kw)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for i in range(2):
    step(cams[i], gts[i], masks[i])
if pipelined:
    step.flush()
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(n):
    step(cams[i % len(cams)], gts[i % len(cams)], masks[i % len(cams)])
if pipelined:
    step.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t
first = step.steps[0] if pipelined else step
what = f"silhouette masks ({100 * float(torch.stack(masks).mean()):.1f} % ones), loss pass A {'sparse' if first._sparse else 'plain'}" \
    if silhouette else "90 %-ones random mask"
print(f"{'pipelined (two captures, results one iteration late)' if pipelined else 'graphed'} s2 step, {what}: "
      f"{n/dt:.1f} it/s, {dt/n*1e3:.3f} ms/it, recaptures {step.recaptures}")
