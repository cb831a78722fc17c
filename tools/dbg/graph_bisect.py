import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from types import SimpleNamespace
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
from ggsplat.render import render
from ggsplat.inner_step import DEFAULT_OPT, DEFAULT_PIPE, registration_step
from ggsplat.loss import fused_photometric_loss
from ggsplat.adam import GraphAdam
stage = sys.argv[1]
W, H = 96, 80
v, f = S.skirt_mesh(24, 40, r_top=0.30, r_bottom=0.5, height=0.8, jitter=2e-3, seed=0)
params = S.skirt_gaussian_params(f.shape[0], sh_degree=0, seed=0)
cams = S.rig_cameras(n_rings=2, n_az=3, radius=2.2, width=W, height=H, f=70.0, seed=0)
cam = cams[0]
for n in ("world_view_transform", "full_proj_transform", "camera_center"):
    setattr(cam, n, getattr(cam, n).cuda())
m = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
m.training_setup(DEFAULT_OPT, is_ff=True)
m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
bg = torch.zeros(3, device="cuda")
gt = torch.rand(3, H, W).cuda(); mask = torch.ones(1, H, W).cuda()

def body():
    if stage == "fwd":
        with torch.no_grad():
            m.update_face_coor()
            return render(cam, m, DEFAULT_PIPE, bg)["render"].sum()
    if stage == "fwdbwd":
        m.update_face_coor()
        l = render(cam, m, DEFAULT_PIPE, bg)["render"].sum(); l.backward(); m.optimizer.zero_grad(); return l
    if stage == "loss":
        m.update_face_coor()
        a, b = fused_photometric_loss(render(cam, m, DEFAULT_PIPE, bg)["render"], gt, mask, 0.2)
        l = a + b; l.backward(); m.optimizer.zero_grad(); return l
    if stage == "step_noopt":
        return registration_step(m, cam, gt, mask, bg, optimizer_step=False, fused_loss=True)["loss"]
    if stage == "step_notrack":
        return registration_step(m, cam, gt, mask, bg, track_densification=False, fused_loss=True)["loss"]
    if stage == "step":
        return registration_step(m, cam, gt, mask, bg, fused_loss=True)["loss"]

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    print("eager", float(body()))
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", float(out), flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed2", float(out), flush=True)
