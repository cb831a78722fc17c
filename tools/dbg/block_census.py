"""Census for a forward / backward whose four 16-lane rows walk DIFFERENT list entries (GPU, torch only):
row g of a wave owns the 4x4 block g of the quadrant being composited and walks only the entries that can reach its block.
Steps of the current kernels = sum over entries of the quadrants in the entry's mask (one 64-pixel body each).
Steps of the row-divergent walk = sum over (tile, quadrant, round of 64 entries) of max over the 4 blocks of the number of
entries reaching the block.  Both with the conservative box test the binning uses (ggs_box_reachable restated in torch) and
with the tight test (some pixel of the block passes the alpha test)."""
import sys, os, math, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras()
sel = [0, 64, 111]
NV = len(sel)
ck = S.stack_cameras([cams[i] for i in sel], device=dev)
with torch.no_grad():
    m.update_face_coor()
    color, radii, depth, alpha, st = R.forward_views(m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None,
        view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
sec = R.bin_sections(st)
n = st.num_rendered
P = m.get_xyz.shape[0]
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
rec = st.geom[:NV * P * 48].view(torch.float32).reshape(NV, P, 12)
cnt = sec["tile_count"].reshape(-1).to(torch.int64)
start = (sec["view_base"].to(torch.int64)[:, None] + sec["tile_offset"].to(torch.int64)).reshape(-1)
words = sec["ids"][:n].to(torch.int64) & 0xffffffff
ne = cnt > 0
st_ne, o = torch.sort(start[ne]); id_ne = torch.nonzero(ne).reshape(-1)[o]
item_of = id_ne[torch.searchsorted(st_ne, torch.arange(n, device=dev), right=True) - 1]       # v * T + t of every entry
pos = torch.arange(n, device=dev) - start[item_of]                                             # position in its list
vv, tt = item_of // T, item_of % T
gid = words & 0x0fffffff
qmask_narrow = (words >> 28) & 15
r = rec[vv, gid]                                                                               # [n, 12]
ox, oy = (tt % gx) * 16, (tt // gx) * 16
LOG2E = 1.4426950408889634

def chunks(n, c=1 << 16):
    for a in range(0, n, c):
        yield slice(a, min(n, a + c))

# tight: per entry a 16-bit mask of 4x4 blocks (bit 4*by + bx) with some pixel passing the alpha test
px = torch.arange(16, device=dev, dtype=torch.float32)
tight = torch.zeros(n, dtype=torch.int64, device=dev)
npass = torch.zeros(n, dtype=torch.int64, device=dev)
for s in chunks(n):
    rr = r[s]
    dx = rr[:, 0, None, None] - (ox[s, None, None].float() + px[None, None, :])               # [c, 1, 16]
    dy = rr[:, 1, None, None] - (oy[s, None, None].float() + px[None, :, None])               # [c, 16, 1]
    power = rr[:, 2, None, None] * dx * dx + rr[:, 4, None, None] * dy * dy + rr[:, 3, None, None] * dx * dy
    al = torch.clamp(rr[:, 5, None, None] * torch.exp2(power), max=0.99)
    ok = (power <= 0) & (al >= 1.0 / 255.0)
    inside = ((ox[s, None, None] + px[None, None, :].long()) < W) & ((oy[s, None, None] + px[None, :, None].long()) < H)
    ok &= inside
    npass[s] = ok.sum((1, 2))
    blk = ok.reshape(-1, 4, 4, 4, 4).any(4).any(2)                                             # [c, by, bx]
    bits = (1 << torch.arange(16, device=dev)).reshape(4, 4)
    tight[s] = (blk.long() * bits).sum((1, 2))

# conservative: ggs_box_reachable on each 4x4 block (csrc/ggs_common.h), from the record fields
def box_reachable(rr, x0, y0, x1, y1):
    mx, my = rr[:, 0], rr[:, 1]
    A, B, C = -rr[:, 2], -0.5 * rr[:, 3], -rr[:, 4]
    op = rr[:, 5]
    tau = torch.where(op > 0, torch.log(255.0 * op.clamp_min(1e-30)), torch.full_like(op, -1.0)) * 1.01 + 0.02
    lim = tau * LOG2E
    left, right, above, below = mx < x0, mx > x1, my < y0, my > y1
    xe = torch.where(right, x1, x0); dxe = xe - mx
    dyv = torch.minimum(y1 - my, torch.maximum(y0 - my, -(B / C) * dxe))
    qv = A * dxe * dxe + 2 * B * dxe * dyv + C * dyv * dyv
    ye = torch.where(below, y1, y0); dye = ye - my
    dxh = torch.minimum(x1 - mx, torch.maximum(x0 - mx, -(B / A) * dye))
    qh = A * dxh * dxh + 2 * B * dxh * dye + C * dye * dye
    ins = ~(left | right | above | below)
    return torch.where(ins, lim > 0, torch.minimum(qv, qh) <= lim)

cons = torch.zeros(n, dtype=torch.int64, device=dev)
for by in range(4):
    for bx in range(4):
        x0 = (ox + 4 * bx).float(); y0 = (oy + 4 * by).float()
        cons |= box_reachable(r, x0, y0, x0 + 3, y0 + 3).long() << (4 * by + bx)
print(f"views {NV}: entries {n} ({n / NV:.0f} per view), pixel passes {int(npass.sum()) / NV:.0f} per view")
print(f"tight blocks not in the conservative mask (must be 0): {int(((tight & ~cons) != 0).sum())}")

QBLOCKS = [[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]       # blocks (bit 4*by+bx) of quadrant q = (qy*2+qx)

def census(mask16, name):
    # per entry: quadrant touched = any of its 4 blocks
    quad = torch.stack([sum(((mask16 >> b) & 1) for b in QBLOCKS[q]).clamp_max(1) for q in range(4)], 1)      # [n, 4]
    cur = int(quad.sum())
    blocks = int(sum(((mask16 >> b) & 1).sum() for b in range(16)))
    rnd = item_of * 64 + pos // 64                                # (item, round) key; lists are < 4096 entries long here
    key, inv = torch.unique(rnd, return_inverse=True)
    new = 0
    for q in range(4):
        c = torch.zeros(len(key), 4, dtype=torch.int64, device=dev)
        for g, b in enumerate(QBLOCKS[q]):
            c[:, g].index_add_(0, inv, (mask16 >> b) & 1)
        new += int(c.max(1).values.sum())
    print(f"{name:>28}: 64-px bodies now {cur / NV:9.0f} per view ({cur / n:.2f} per entry), touched 4x4 blocks {blocks / NV:9.0f} "
          f"({blocks / n:.2f} per entry, {blocks / 4 / NV:.0f} ideal steps), row-divergent steps {new / NV:9.0f}  ->  x{cur / new:.2f} fewer steps")

census(cons, "conservative box test")
census(tight, "tight (a pixel passes)")
# what the kernels do today: the binning's quadrant masks (forward) and the forward-narrowed masks (backward)
print(f"narrowed quadrant masks (what the backward walks): {int(sum(((qmask_narrow >> q) & 1).sum() for q in range(4))) / NV:.0f} bodies per view")
