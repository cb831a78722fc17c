"""Unprofiled host time of the pieces of the eager per-view loop (render() + backward, bench: per_view_loop_views_per_sec) on a GPU
box: perf_counter wrappers around the functions of this repo that a call passes through (~0.3 us each; cProfile inflates small
Python functions several-fold).  Nested: render > rasterize.forward > forward_views > {_prep_forward, ggs_forward_spec, header wait}."""
import os, sys, time, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace
from ggsplat import synthetic as S, rasterizer as R, mesh_gaussian_model as MG, render as RD, _lib

acc = collections.defaultdict(float); cnt = collections.Counter()


def timed(f, k):
    def g(*a, **kw):
        t0 = time.perf_counter(); r = f(*a, **kw); acc[k] += time.perf_counter() - t0; cnt[k] += 1; return r
    return g


def wrap_cls(cls, name):
    for meth in ("forward", "backward"):
        setattr(cls, meth, staticmethod(timed(getattr(cls, meth), f"{name}.{meth}")))


wrap_cls(R._RasterizeGaussians, "node rasterize"); wrap_cls(MG._MeshBind, "node mesh_bind")
for mod, names in ((R, ("forward_views", "backward_views", "_prep_forward", "_workspace_sizes", "_tile_count_offset", "new_grads")),
                   (RD, ("_settings",)), (MG, ("mesh_bind",))):
    for n in names:
        if hasattr(mod, n):
            setattr(mod, n, timed(getattr(mod, n), f"{mod.__name__.split('.')[-1]}.{n}"))
MG.MeshGaussianModel._bind = timed(MG.MeshGaussianModel._bind, "model._bind (x3 per render)")
torch.cuda.Event.synchronize = timed(torch.cuda.Event.synchronize, "Event.synchronize (header wait)")
L = _lib.lib()


class LibProxy:         # times the two C calls of the rasterizer
    def __getattr__(self, n):
        f = getattr(L, n)
        if n in ("ggs_forward_spec", "ggs_backward", "ggs_mesh_bind_forward", "ggs_mesh_bind_backward"):
            f = timed(f, f"C {n}")
            setattr(self, n, f)
        return f


proxy = LibProxy()
R.lib = lambda: proxy
MG.lib = lambda: proxy
render = timed(RD.render, "render()")
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MG.MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)[:32]
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); w = torch.randn(3, H, W, device=dev)
bwd = timed(lambda t: t.backward(), "loss.backward() (engine + our backward nodes)")
for rep in range(4):
    acc.clear(); cnt.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for c in cams:
        m.update_face_coor()
        pkg = render(c, m, pipe, bg)
        bwd((pkg["render"] * w).sum())
        for q in m.parameters():
            q.grad = None
    t1 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"per-view loop: {dt / len(cams) * 1e6:.0f} us per iteration ({len(cams) / dt:.0f} / s), host issue {(t1 - t0) / len(cams) * 1e6:.0f} us")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k:52s} {acc[k] / len(cams) * 1e6:7.1f} us per iteration  ({cnt[k] // len(cams)} calls)")
