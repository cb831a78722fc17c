"""fwd+bwd time per launch of V views (config 2) -- run with GGS_QUAD_ITEMS=0 (throughput kernels) / 1000000 (latency)."""
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
out = []
for V in (1, 2, 4, 6, 8, 10, 12, 16):
    ck = S.stack_cameras([cams[(7 * i) % 160] for i in range(V)], device=dev)
    dL = torch.randn(V, 3, H, W, device=dev)
    def run():
        c, r, d, a, st = R.forward_views(*inp, view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"],
                                         bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
        R.backward_views(st, dL, want_means2D=False)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); out.append((V, (time.perf_counter() - t) / 10 * 1e3))
print(os.environ.get("GGS_QUAD_ITEMS", "default"), " ".join(f"V={V}:{ms:.3f}ms" for V, ms in out))
