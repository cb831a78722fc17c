"""Replica mode (ggsplat.inner_step.ReplicaRegistrationSteps) at config 2: R independent s2 registrations, one optimiser step per
view each, replayed side by side on R streams of one GPU.  Prints the aggregate iteration rate for R = 1, 2, 4, 8 (two passes each),
next to the sequential and the pipelined captured step.  `python tools/dbg/replica_rate.py [N_ITERATIONS]`; run it under
`rocprofv3 --kernel-trace` with --trace R to get the overlap of one R (tools/overlap_summary.py)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from types import SimpleNamespace
from ggsplat import synthetic as S
from ggsplat.adam import GraphAdam
from ggsplat.inner_step import DEFAULT_OPT, GraphedRegistrationStep, PipelinedRegistrationStep, ReplicaRegistrationSteps
from ggsplat.mesh_gaussian_model import MeshGaussianModel
from ggsplat.render import render
dev, W, H = "cuda", 1920, 1080
trace = None
if "--trace" in sys.argv:
    i = sys.argv.index("--trace"); trace = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
v, f = S.skirt_mesh()


def model(seed):
    m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], 0, seed=seed), 0, device=dev)
    m.training_setup(DEFAULT_OPT, is_ff=True)
    m.optimizer = GraphAdam(m.optimizer.param_groups, lr=0.0, eps=1e-15)
    return m


cams = S.rig_cameras(device=dev)[:16]
for c in cams:
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(c, name, getattr(c, name).to(dev))
bg = torch.zeros(3, device=dev)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
m0 = model(0)
with torch.no_grad():
    m0.update_face_coor()
    gts = [(render(c, m0, pipe, bg)["render"] + 0.02 * torch.randn(3, H, W, device=dev)).clamp(0, 1).contiguous() for c in cams]
mask = (torch.rand(1, H, W, device=dev) > 0.1).float()


def timed(fn, iters):
    fn(2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


if trace is None:
    seq = GraphedRegistrationStep(model(0), W, H, bg)
    dt = timed(lambda k: [seq(cams[i % 16], gts[i % 16], mask) for i in range(k)], n)
    print(f"sequential captured step: {1 / dt:.0f} it/s ({dt * 1e6:.1f} us per iteration)")
    pip = PipelinedRegistrationStep(model(0), W, H, bg)
    dt = timed(lambda k: ([pip(cams[i % 16], gts[i % 16], mask) for i in range(k)], pip.flush()), n)
    print(f"pipelined captured step:  {1 / dt:.0f} it/s ({dt * 1e6:.1f} us per iteration)")
    del seq, pip
for R_n in ((1, 2, 4, 8) if trace is None else (trace,)):
    reps = ReplicaRegistrationSteps([model(r) for r in range(R_n)], W, H, bg)

    def rounds(k):
        for i in range(k):
            idx = [(i + 5 * r) % 16 for r in range(R_n)]
            reps([cams[j] for j in idx], [gts[j] for j in idx], [mask] * R_n)
    for _ in range(2):
        dt = timed(rounds, n)
        print(f"R = {R_n} replicas: {R_n / dt:.0f} it/s aggregate ({dt * 1e6:.1f} us per round of {R_n} iterations, {dt / R_n * 1e6:.1f} us per iteration), recaptures {reps.recaptures}")
    del reps
