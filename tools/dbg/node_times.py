"""Host time inside this repo's autograd nodes during the eager s2 step (GPU box): wraps forward / backward of the four custom
Functions with perf_counter (the backward methods run on PyTorch's engine thread, where cProfile of the main thread is blind)."""
import os, sys, time, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace
from ggsplat import synthetic as S, rasterizer as R, loss as LO, inner_step as IS, mesh_gaussian_model as MG
from ggsplat.inner_step import DEFAULT_OPT, registration_step
from ggsplat.render import render

acc = collections.defaultdict(float); cnt = collections.Counter()


def wrap(cls, name):
    for meth in ("forward", "backward"):
        f = getattr(cls, meth)
        def g(*a, _f=f, _k=f"{name}.{meth}", **kw):
            t0 = time.perf_counter(); r = _f(*a, **kw); acc[_k] += time.perf_counter() - t0; cnt[_k] += 1; return r
        setattr(cls, meth, staticmethod(g))


wrap(R._RasterizeGaussians, "rasterize"); wrap(MG._MeshBind, "mesh_bind"); wrap(LO._FusedPhotometric, "photometric"); wrap(IS._HingeRegularisers, "hinges")
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MG.MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)[:32]
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    m.update_face_coor()
    gts = [(render(c, m, pipe, bg)["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp_(0, 1).contiguous() for c in cams]
mask = (torch.rand(1, H, W, device=dev) > 0.1).float()
m.training_setup(DEFAULT_OPT, is_ff=True)
for rep in range(3):
    acc.clear(); cnt.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for c, gt in zip(cams, gts):
        registration_step(m, c, gt, mask, bg, fused_loss=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"eager step: {dt / len(cams) * 1e6:.0f} us per iteration")
for k in sorted(acc):
    print(f"  {k:24s} {acc[k] / cnt[k] * 1e6:7.1f} us  x {cnt[k] // len(cams)} per iteration")
print(f"  sum of the eight          {sum(acc.values()) / len(cams) * 1e6:7.1f} us")
