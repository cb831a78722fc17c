"""Host-side profile of the per-view drop-in path (render() + backward), GPU box."""
import sys, os, time, torch, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace
from ggsplat import synthetic as S
from ggsplat.mesh_gaussian_model import MeshGaussianModel
from ggsplat.render import render
dev = "cuda"
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)[:32]
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
w = torch.randn(3, 1080, 1920, device=dev)
def loop():
    for c in cams:
        m.update_face_coor()
        pkg = render(c, m, pipe, bg)
        (pkg["render"] * w).sum().backward()
        for q in m.parameters(): q.grad = None
loop(); torch.cuda.synchronize()
t = time.perf_counter(); loop(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"per-view loop: {len(cams)/dt:.1f} views/s, {dt/len(cams)*1e3:.3f} ms/view")
pr = cProfile.Profile(); pr.enable(); loop(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
