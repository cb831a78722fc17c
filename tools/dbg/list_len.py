import sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras(device=dev)
with torch.no_grad():
    for ci in (0, 40, 80, 159):
        c = cams[ci]
        view = c.world_view_transform.to(dev).reshape(1, 16); proj = c.full_proj_transform.to(dev).reshape(1, 16)
        campos = c.camera_center.to(dev).reshape(1, 3)
        tanfov = torch.tensor([[math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5)]], device=dev)
        color, radii, depth, alpha, st = R.forward_views(m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None,
            view=view, proj=proj, campos=campos, tanfov=tanfov, bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
        tc = R.bin_sections(st)["tile_count"][0].float()
        nz = tc[tc > 0]
        q = torch.quantile(nz, torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
        print(f"cam {ci}: N={int(tc.sum())} nonempty={nz.numel()} mean={nz.mean():.1f} p50={q[0]:.0f} p90={q[1]:.0f} p99={q[2]:.0f} p99.9={q[3]:.0f} max={int(tc.max())}")
