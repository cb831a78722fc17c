cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_loss.py tests/test_gpu_graph_step.py tests/test_gpu_inner_step.py -q -x 2>&1 | tail -3
cat > /tmp/loss_time.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "gaussian-garments_amd"))
from ggsplat.loss import fused_photometric_loss
for V in (1, 8, 32):
    H, W = 1080, 1920
    img = torch.rand(V, 3, H, W, device="cuda"); gt = torch.rand(V, 3, H, W, device="cuda")
    mask = (torch.rand(V, 1, H, W, device="cuda") > 0.2).float()
    def run():
        z = img.clone().requires_grad_(True)
        a, b = fused_photometric_loss(z, gt, mask, 0.2)
        (a + b).sum().backward()
        return z.grad
    g = run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 10)
    print(f"V={V}: {best*1e3:.3f} ms  ({best/V*1e6:.1f} us/view)  grad checksum {float(g.double().sum()):.9e} {float(g.double().abs().sum()):.9e}")
PY
for s in 1; do echo "GGS_LOSS_STREAM=$s"; GGS_LOSS_STREAM=$s python /tmp/loss_time.py; done
cd /tmp; for s in 1; do GGS_LOSS_STREAM=$s rocprofv3 --kernel-trace --stats -d /tmp/lp$s -o l -- python /tmp/loss_time.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/lp$s -name "*.db" | head -1) 2>/dev/null | grep "ggs_k_loss" | cut -c1-120; done
