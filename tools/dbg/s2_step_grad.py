import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
from ggsplat import synthetic as S
from ggsplat.inner_step import DEFAULT_OPT, registration_step
from ggsplat.mesh_gaussian_model import MeshGaussianModel
import test_gpu_fullsize_steps as T
from oracle import host_oracle as HO

v, f = S.skirt_mesh(); P = f.shape[0]; cams = S.rig_cameras()
opt = SimpleNamespace(**{**vars(DEFAULT_OPT), "threshold_xyz": 0.002, "threshold_scale": 0.5})
params = S.skirt_gaussian_params(P, sh_degree=0)
params["_xyz"] = torch.randn(P, 3, generator=torch.Generator().manual_seed(31)) * 0.05
gt, mask = T._images(32); bg = torch.zeros(3); cam = T._cam_to(cams[13], "cuda")
model = MeshGaussianModel.from_tensors(v, f, params, sh_degree=0, device="cuda")
model.training_setup(opt, is_ff=True)
out = registration_step(model, cam, gt.cuda(), mask.cuda(), bg.cuda(), opt=opt, optimizer_step=False, fused_loss=True)
pkg = out["render_pkg"]
leaf = {n: params[n].clone().requires_grad_(True) for n in T.NAMES}
mv = v.clone().requires_grad_(True)
xyz, scaling, rot = HO.mesh_bind(mv, f, params["binding"], leaf["_xyz"], leaf["_scaling"], leaf["_rotation"])
shs = torch.cat((leaf["_features_dc"], leaf["_features_rest"]), 1); opacity = torch.sigmoid(leaf["_opacity"])
co, img = T._c_render(cam, xyz, scaling, rot, opacity, shs, 0, bg)
vis = torch.from_numpy(co.radii > 0)
l_xyz = F.relu(leaf["_xyz"][vis].norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz
l_sc = F.relu(torch.exp(leaf["_scaling"][vis]) - opt.threshold_scale).norm(dim=1).mean() * opt.lambda_scale
(l_xyz + l_sc).backward(retain_graph=True)
hinge = leaf["_xyz"].grad.clone()
dimg = T._gpu_loss_grad(pkg, gt, mask, opt.lambda_dssim)
g = co.backward(dimg.numpy())
torch.autograd.backward([xyz, scaling, rot, opacity, shs], [torch.from_numpy(g[k]).reshape(t.shape) for k, t in zip(("means3D", "scales", "rotations", "opacities", "shs"), (xyz, scaling, rot, opacity, shs))])
ref = leaf["_xyz"].grad; photo = ref - hinge
gpu = model._xyz.grad.cpu()
# GPU hinge alone
px = params["_xyz"].clone().cuda().requires_grad_(True)
visf = torch.ones(P, device="cuda")
(((F.relu(px.norm(dim=1) - opt.threshold_xyz) * visf).sum() / visf.sum()) * opt.lambda_xyz).backward()
print("sum|hinge|", float(hinge.abs().sum()), "sum|photo|", float(photo.abs().sum()), "sum|err|", float((gpu - ref).abs().sum()))
print("hinge gpu vs cpu rel", float((px.grad.cpu() - hinge).abs().sum() / hinge.abs().sum()))
e = (gpu - ref).abs()
print("err of (gpu - gpu_hinge) vs photo:", float(((gpu - px.grad.cpu()) - photo).abs().sum() / photo.abs().sum()))
idx = torch.topk(e.sum(1), 5).indices
print("top err rows", idx.tolist(), e[idx].tolist(), "ref", ref[idx].tolist(), "radii", co.radii[idx.numpy()].tolist())
print("quantiles of row err / row scale:", torch.quantile((e.sum(1) / (ref.abs().sum(1) + 1e-12)), torch.tensor([0.5, 0.9, 0.99, 0.999])).tolist())
print("means3D grad gpu chain? compare viewspace:", float((pkg["viewspace_points"].grad.cpu() - torch.from_numpy(g["means2D"])).abs().sum() / abs(g["means2D"]).sum()))
