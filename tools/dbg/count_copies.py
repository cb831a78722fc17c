import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S
from ggsplat.adam import GraphAdam
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep
from ggsplat.mesh_gaussian_model import MeshGaussianModel
mode = sys.argv[1]
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev); m.training_setup(DEFAULT_OPT, is_ff=True)
m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
cams = S.rig_cameras(device=dev)[:4]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev); gt = torch.rand(3, H, W, device=dev); mask = torch.ones(1, H, W, device=dev)
step = GraphedRegistrationStep(m, W, H, bg)
step(cams[0], gt, mask); torch.cuda.synchronize()
N = 100
for i in range(N):
    if mode == "load":
        step._load(cams[i % 4], gt, mask)
    elif mode == "replay":
        step.graph.replay()
    elif mode == "post":
        step._hdr_host.copy_(step._hdr_dev, non_blocking=True); step._stats_host.copy_(step._stats, non_blocking=True)
    torch.cuda.synchronize()
