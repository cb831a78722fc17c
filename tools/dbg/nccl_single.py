"""One-rank RCCL process group next to a captured compute graph: exercises init_process_group("nccl", device_id),
all_reduce of the flat gradient bucket, barrier and graph capture / replay while the RCCL watchdog thread is alive."""
import os, sys, time, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from ggsplat import batch, synthetic as S
from ggsplat.dist import all_reduce_grads
from ggsplat.mesh_gaussian_model import MeshGaussianModel
W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
model = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.stack_cameras(S.rig_cameras()[:20], device=dev)
bg = torch.zeros(3, device=dev); w_img = torch.randn(3, H, W, device=dev)
dL = w_img.unsqueeze(0).expand(20, 3, H, W).contiguous()
plist = model.parameters()
def compute():
    for q in plist: q.grad = None
    model.update_face_coor()
    xyz, sc, rot, op, shs = model.get_xyz, model.get_scaling, model.get_rotation, model.get_opacity, model.get_features
    inputs = dict(means3D=xyz.detach(), scales=sc.detach(), rotations=rot.detach(), opacities=op.detach(), shs=shs.detach())
    gr = batch.fwd_bwd_views(inputs, cams, bg=bg, W=W, H=H, sh_degree=0, chunk=20, dL_dcolor_fn=lambda a, b, c: dL[:b - a])
    torch.autograd.backward([xyz, sc, rot, op, shs], [gr["means3D"], gr["scales"], gr["rotations"], gr["opacities"], gr["shs"]])
    return [q.grad if q.grad is not None else torch.zeros_like(q) for q in plist]
g0 = compute(); all_reduce_grads(g0, 20); dist.barrier(); torch.cuda.synchronize()
ref = [t.clone() for t in g0]
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph, capture_error_mode="thread_local"):
    gs = compute()
for _ in range(3):
    gph.replay(); all_reduce_grads(gs, 20)
dist.barrier(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    gph.replay(); all_reduce_grads(gs, 20)
dist.barrier(); torch.cuda.synchronize()
dt = time.perf_counter() - t
err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(gs, ref) if a.numel())
print(f"rccl single-rank ok: {20 * 20 / dt:.0f} views/s with graph + all_reduce, max rel diff vs eager {err:.2e}")
dist.destroy_process_group()
