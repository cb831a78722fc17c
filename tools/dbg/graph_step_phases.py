"""Where the host time of a graph-replayed s2 iteration goes: per-phase wall clock (load / replay / read-back + sync)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S
from ggsplat.adam import GraphAdam
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
m.training_setup(DEFAULT_OPT, is_ff=True)
m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
cams = S.rig_cameras(device=dev)[:16]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev)
# target = the model's own first render + noise (against a random image the Gaussians grow and the iterations get slower)
from types import SimpleNamespace
from ggsplat.render import render
with torch.no_grad():
    m.update_face_coor()
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    gts = [(render(c, m, pipe, bg)["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp_(0, 1).contiguous() for c in cams]
mask = (torch.rand(1, H, W, device=dev) > 0.1).float()
step = GraphedRegistrationStep(m, W, H, bg)
for c, g_ in zip(cams, gts):
    step(c, g_, mask)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128      # (the synthetic schedule has no lr decay: keep the run short)
T = [0.0, 0.0, 0.0]; per = []
st = torch.cuda.current_stream()
for i in range(n):
    c = cams[i % 16]
    t0 = time.perf_counter(); step._load(c, gts[i % 16], mask)
    t1 = time.perf_counter(); step.graph.replay()
    t2 = time.perf_counter()
    if not step._out_map:
        step._out_host.copy_(step._out, non_blocking=True)
    st.synchronize(); d = step._losses_from_stats()
    t3 = time.perf_counter()
    T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2; per.append(t3 - t0)
print(f"n={n}: load {T[0]/n*1e6:.1f} us, replay call {T[1]/n*1e6:.1f} us, sync + result {T[2]/n*1e6:.1f} us, total {sum(T)/n*1e6:.1f} us/it "
      f"(parameter / result blocks in mapped pinned memory: {bool(step._blk_map)} / {bool(step._out_map)})")
import statistics
for a in range(0, n, 64):
    seg = per[a:a + 64]
    print(f"  it {a:4d}..{a+len(seg)-1:4d}: median {statistics.median(seg)*1e6:.0f} us, max {max(seg)*1e6:.0f} us")
m2 = 64
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(m2):
    step.graph.replay()
torch.cuda.synchronize()
print(f"replay back to back, one sync at the end (the GPU's own pace): {(time.perf_counter()-t0)/m2*1e6:.1f} us/it")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(m2):
    step(cams[i % 16], gts[i % 16], mask)
print(f"step(): {(time.perf_counter()-t0)/m2*1e6:.1f} us/it")
