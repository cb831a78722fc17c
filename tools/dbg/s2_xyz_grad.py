import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ggsplat import synthetic as S
from ggsplat.mesh_gaussian_model import mesh_bind
from oracle import host_oracle as HO

def rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().sum() / (b.abs().sum() + 1e-30))

v, f = S.skirt_mesh(); P = f.shape[0]
params = S.skirt_gaussian_params(P, sh_degree=0)
params["_xyz"] = torch.randn(P, 3, generator=torch.Generator().manual_seed(31)) * 0.05
g = torch.Generator().manual_seed(3)
for name, gx in (("white", torch.randn(P, 3, generator=g)), ("aligned with normal", None)):
    leaf = {k: params[k].clone().requires_grad_(True) for k in ("_xyz", "_scaling", "_rotation")}
    mv = v.clone().requires_grad_(True)
    xyz, sc, rot = HO.mesh_bind(mv, f, params["binding"], leaf["_xyz"], leaf["_scaling"], leaf["_rotation"])
    if gx is None:
        R, _ = HO.compute_face_orientation(v, f)
        gx = R[:, :, 1] * torch.randn(P, 1, generator=g) + 0.01 * torch.randn(P, 3, generator=g)
    gs, gr = torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)
    torch.autograd.backward([xyz, sc, rot], [gx, gs, gr])
    dl = {k: params[k].clone().cuda().requires_grad_(True) for k in ("_xyz", "_scaling", "_rotation")}
    dv = v.clone().cuda().requires_grad_(True)
    x2, s2, r2 = mesh_bind(dv, f.cuda(), params["binding"].cuda(), dl["_xyz"], dl["_scaling"], dl["_rotation"])
    torch.autograd.backward([x2, s2, r2], [gx.cuda(), gs.cuda(), gr.cuda()])
    print(name, "fwd xyz", rel(x2, xyz), "| d_xyz", rel(dl["_xyz"].grad, leaf["_xyz"].grad), "d_scaling", rel(dl["_scaling"].grad, leaf["_scaling"].grad),
          "d_rot", rel(dl["_rotation"].grad, leaf["_rotation"].grad), "d_verts", rel(dv.grad, mv.grad))
    e = (dl["_xyz"].grad.cpu() - leaf["_xyz"].grad).abs()
    print("   per-component |err| sums", e.sum(0).tolist(), "ref sums", leaf["_xyz"].grad.abs().sum(0).tolist())
