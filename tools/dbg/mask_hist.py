import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)
with torch.no_grad():
    inp = dict(means3D=m.get_xyz, opacities=m.get_opacity, shs=m.get_features, scales=m.get_scaling, rotations=m.get_rotation)
    view = torch.stack([c.world_view_transform.to(dev).reshape(16) for c in cams[:8]])
    proj = torch.stack([c.full_proj_transform.to(dev).reshape(16) for c in cams[:8]])
    campos = torch.stack([c.camera_center.to(dev) for c in cams[:8]])
    import math
    tanfov = torch.tensor([[math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5)] for c in cams[:8]], device=dev)
    color, radii, depth, alpha, st = R.forward_views(inp["means3D"], inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"], None,
        view=view, proj=proj, campos=campos, tanfov=tanfov, bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
    sec = R.bin_sections(st)
    n = st.num_rendered
    ids = sec["ids"][:n].to(torch.int64) & 0xffffffff
    keys = sec["keys"][:n]
    for name, mk in (("forward (scatter mask, from keys)", (keys >> 28) & 15), ("backward (narrowed, from ids)", (ids >> 28) & 15)):
        h = torch.bincount(mk, minlength=16).float()
        tot = h.sum()
        print(name, "entries", int(tot))
        pc = [bin(i).count("1") for i in range(16)]
        for q in range(5):
            print(f"  popcount {q}: {sum(h[i] for i in range(16) if pc[i]==q)/tot*100:.1f}%")
        both01 = sum(h[i] for i in range(16) if (i & 3) == 3) / tot
        both23 = sum(h[i] for i in range(16) if (i & 12) == 12) / tot
        both02 = sum(h[i] for i in range(16) if (i & 5) == 5) / tot
        both13 = sum(h[i] for i in range(16) if (i & 10) == 10) / tot
        qwork = sum(h[i] * pc[i] for i in range(16)) / tot
        print(f"  mean quadrants {qwork:.2f}; pairs both active: (0,1) {both01*100:.1f}% (2,3) {both23*100:.1f}% | (0,2) {both02*100:.1f}% (1,3) {both13*100:.1f}%")
