import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import stylenet as SN
def rel(a, b): return float((a.double() - b.double()).abs().sum() / (b.double().abs().sum() + 1e-30))
for dt in (torch.float32, torch.float64):
    ch = {k: min(v, 24) for k, v in SN.CHANNELS.items()}
    torch.manual_seed(5)
    hip = SN.StyleUNetLite(64, 4, 7, 32, impl="hip", channels=ch).cuda().to(dt)
    nat = SN.StyleUNetLite(64, 4, 7, 32, impl="native", channels=ch).cuda().to(dt)
    nat.load_state_dict(copy.deepcopy(hip.state_dict()))
    nat2 = copy.deepcopy(nat)
    g = torch.Generator().manual_seed(6)
    cond, style = torch.randn(2, 4, 64, 64, generator=g).cuda().to(dt), torch.randn(2, 32, generator=g).cuda().to(dt)
    w = torch.randn(2, 7, 64, 64, generator=g).cuda().to(dt)
    outs = []
    for net in (hip, nat, nat2):
        out = net(cond, style); (out * w).sum().backward(); outs.append(out.detach())
    print(dt, "out hip/nat", rel(outs[0], outs[1]), "nat/nat", rel(outs[2], outs[1]))
    for (n, p), (_, q), (_, r) in zip(hip.named_parameters(), nat.named_parameters(), nat2.named_parameters()):
        e, e2 = rel(p.grad, q.grad), rel(r.grad, q.grad)
        if e > 1e-5 or dt is torch.float64: print(f"  {n:28s} hip-vs-native {e:.2e}   native-vs-native {e2:.2e}")
