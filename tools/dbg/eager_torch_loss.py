import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S
from ggsplat.inner_step import DEFAULT_OPT, registration_step
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev); m.training_setup(DEFAULT_OPT, is_ff=True)
cams = S.rig_cameras(device=dev)[:16]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev); gt = torch.rand(3, H, W, device=dev); mask = (torch.rand(1, H, W, device=dev) > 0.1).float()
for fused in (False, True):
    for c in cams[:2]: registration_step(m, c, gt.clone(), mask, bg, fused_loss=fused)
    torch.cuda.synchronize(); t = time.perf_counter()
    for c in cams: registration_step(m, c, gt.clone(), mask, bg, fused_loss=fused)
    torch.cuda.synchronize(); print("fused_loss", fused, round(len(cams) / (time.perf_counter() - t), 1), "it/s")
