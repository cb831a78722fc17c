#!/usr/bin/env python
"""Forward-only kernel times (HIP events of ggs_profile_*) at config 2 for the library named by GGS_LIB_PATH: what-if builds
of the forward (no empty-tile stores / every tile empty / non-temporal stores) price the background stores against the
compositing.  Never calls the backward (a what-if forward leaves the workspace undefined).  Usage: time_fwd.py [views] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
import torch  # noqa: E402
from ggsplat import _lib, rasterizer as R, synthetic as S  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
W, H = 1920, 1080
verts, faces = S.skirt_mesh(200, 250)
model = MeshGaussianModel.from_tensors(verts, faces, S.skirt_gaussian_params(faces.shape[0], sh_degree=0), sh_degree=0, device=dev)
cams = S.stack_cameras(S.rig_cameras(n_rings=max(1, V // 32), n_az=min(32, V), width=W, height=H, f=1500.0)[:V], device=dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    inp = dict(means3D=model.get_xyz, scales=model.get_scaling, rotations=model.get_rotation, opacities=model.get_opacity,
               shs=model.get_features)
L = _lib.lib()
names = ["preprocess", "scan", "scatter", "sort", "render_fwd"]
acc = [0.0] * 5
for i in range(reps + 1):
    L.ggs_profile_enable(1 if i else 0)
    out = R.forward_views(inp["means3D"], inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"], None,
                          view=cams["view"], proj=cams["proj"], campos=cams["campos"], tanfov=cams["tanfov"], bg=bg, W=W, H=H,
                          sh_degree=0)
    torch.cuda.synchronize()
    if i:
        buf = (C.c_float * 8)()
        L.ggs_profile_read(buf, 8)
        acc = [a + b for a, b in zip(acc, list(buf)[:5])]
    del out
L.ggs_profile_enable(0)
print(os.path.basename(os.environ.get("GGS_LIB_PATH", "product")), f"V={V}:",
      "  ".join(f"{n} {a / reps / V * 1e3:.2f}" for n, a in zip(names, acc)), "us/view")
