import sys, os, time, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras()
with torch.no_grad():
    inp = (m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None)
ck = S.stack_cameras([cams[3]], device=dev); dL = torch.randn(1, 3, H, W, device=dev); bg = torch.zeros(3, device=dev)
N = 200
tf = tb = 0.0
for i in range(N + 20):
    t0 = time.perf_counter()
    c, r, d, a, st = R.forward_views(*inp, view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=bg, W=W, H=H, sh_degree=0)
    t1 = time.perf_counter()
    g = R.backward_views(st, dL, want_means2D=True)
    t2 = time.perf_counter()
    if i >= 20: tf += t1 - t0; tb += t2 - t1
    if i % 8 == 7: torch.cuda.synchronize()
torch.cuda.synchronize()
print(f"host time per call: forward_views {tf/N*1e6:.0f} us (includes its header sync), backward_views {tb/N*1e6:.0f} us")
