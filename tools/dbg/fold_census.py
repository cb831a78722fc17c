"""Census for the two re-mappings VERDICT r3 #1 asks to build (GPU, torch only; config 2, three views):
 (a) larger tiles: (splat, tile) entries per view with 16x16 tiles (today), 32x16 and 32x32 tiles (an entry of a larger tile =
     the union of the 16x16 entries it covers);
 (b) one pass per entry by folding lanes mod 8: an entry whose pixels that pass the alpha test inside the tile fit an 8x8 window
     can be evaluated in ONE 64-lane pass (lane = (x & 7, y & 7)); reported: the distribution of quadrant passes per entry today
     (forward: the binning's conservative masks; backward: the forward-narrowed masks) and the passes left if every eligible
     multi-quadrant entry took one folded pass -- with the tight window (pixels that pass) and with the conservative one
     (the record's alpha AABB clipped to the tile, what a kernel can know without testing pixels)."""
import sys, os, torch
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
reci = st.geom[:NV * P * 48].view(torch.int32).reshape(NV, P, 12)
cnt = sec["tile_count"].reshape(-1).to(torch.int64)
start = (sec["view_base"].to(torch.int64)[:, None] + sec["tile_offset"].to(torch.int64)).reshape(-1)
words = sec["ids"][:n].to(torch.int64) & 0xffffffff
ne = cnt > 0
st_ne, o = torch.sort(start[ne]); id_ne = torch.nonzero(ne).reshape(-1)[o]
item_of = id_ne[torch.searchsorted(st_ne, torch.arange(n, device=dev), right=True) - 1]
vv, tt = item_of // T, item_of % T
gid = words & 0x0fffffff
bwd_mask = (words >> 28) & 15
r = rec[vv, gid]
ri = reci[vv, gid]
tx, ty = tt % gx, tt // gx
ox, oy = tx * 16, ty * 16
px = torch.arange(16, device=dev, dtype=torch.float32)


def chunks(n, c=1 << 16):
    for a in range(0, n, c):
        yield slice(a, min(n, a + c))


# tight: pixels of the tile that pass the alpha test -> per quadrant any, and the window (bounding box) of all of them
qany = torch.zeros(n, 4, dtype=torch.bool, device=dev)
wx = torch.zeros(n, dtype=torch.int64, device=dev); wy = torch.zeros(n, dtype=torch.int64, device=dev)
npass = torch.zeros(n, dtype=torch.int64, device=dev)
idx = torch.arange(16, device=dev)
for s in chunks(n):
    rr = r[s]
    dx = rr[:, 0, None, None] - (ox[s, None, None].float() + px[None, None, :])
    dy = rr[:, 1, None, None] - (oy[s, None, None].float() + px[None, :, None])
    power = rr[:, 2, None, None] * dx * dx + rr[:, 4, None, None] * dy * dy + rr[:, 3, None, None] * dx * dy
    al = torch.clamp(rr[:, 5, None, None] * torch.exp2(power), max=0.99)
    ok = (power <= 0) & (al >= 1.0 / 255.0)
    ok &= ((ox[s, None, None] + px[None, None, :].long()) < W) & ((oy[s, None, None] + px[None, :, None].long()) < H)
    npass[s] = ok.sum((1, 2))
    q = ok.reshape(-1, 2, 8, 2, 8).any(4).any(2)            # [c, qy, qx]
    qany[s] = q.reshape(-1, 4)
    cols = ok.any(1); rows = ok.any(2)                       # [c, 16]
    xmin = torch.where(cols, idx, 99).min(1).values; xmax = torch.where(cols, idx, -1).max(1).values
    ymin = torch.where(rows, idx, 99).min(1).values; ymax = torch.where(rows, idx, -1).max(1).values
    wx[s] = xmax - xmin + 1; wy[s] = ymax - ymin + 1


# conservative window: the record's alpha AABB (int16 min | max << 16) clipped to the tile
def lo(w): return ((w << 48) >> 48)
def hi(w): return (w >> 16)


bbx, bby = ri[:, 10].to(torch.int64), ri[:, 11].to(torch.int64)
cx0 = torch.maximum(lo(bbx), ox); cx1 = torch.minimum(hi(bbx), ox + 15)
cy0 = torch.maximum(lo(bby), oy); cy1 = torch.minimum(hi(bby), oy + 15)
cwx, cwy = cx1 - cx0 + 1, cy1 - cy0 + 1

nq_tight = qany.sum(1)
nq_bwd = sum(((bwd_mask >> q) & 1) for q in range(4))
# forward masks today = conservative box test per quadrant (csrc/ggs_common.h ggs_quad_mask), restated
LOG2E = 1.4426950408889634


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


nq_fwd = torch.zeros(n, dtype=torch.int64, device=dev)
for q in range(4):
    x0 = (ox + 8 * (q & 1)).float(); y0 = (oy + 8 * (q >> 1)).float()
    hit = box_reachable(r, x0, y0, x0 + 7, y0 + 7)
    hit &= (lo(bbx) <= x0 + 7) & (hi(bbx) >= x0) & (lo(bby) <= y0 + 7) & (hi(bby) >= y0)
    nq_fwd += hit.long()
nq_fwd = nq_fwd.clamp_min(1)


def dist(x, name):
    h = torch.bincount(x.clamp(0, 4), minlength=5).tolist()
    print(f"{name:>44}: " + "  ".join(f"{k}q {c / n * 100:5.1f}%" for k, c in enumerate(h)) + f"   mean {float(x.float().mean()):.3f} passes / entry")


print(f"views {NV}: entries {n} ({n / NV:.0f} per view), alpha-test passes {int(npass.sum()) / NV:.0f} px per view ({int(npass.sum()) / n:.1f} per entry)")
dist(nq_fwd, "forward today (binning's quadrant masks)")
dist(nq_tight, "tight (a pixel of the quadrant passes)")
dist(nq_bwd, "backward today (forward-narrowed masks)")


def folded(nq, okw, name):
    multi = nq >= 2
    el = multi & okw
    passes = torch.where(el, torch.ones_like(nq), nq)
    by = {k: int(((nq == k) & el).sum()) for k in (2, 3, 4)}
    print(f"{name:>44}: multi-quadrant entries {int(multi.sum()) / n * 100:5.1f}%, of which eligible {int(el.sum()) / max(1, int(multi.sum())) * 100:5.1f}% "
          f"(2q {by[2] / n * 100:.1f}% 3q {by[3] / n * 100:.1f}% 4q {by[4] / n * 100:.1f}% of all entries); "
          f"passes / entry {float(nq.float().mean()):.3f} -> {float(passes.float().mean()):.3f}")


tight_ok = (wx <= 8) & (wy <= 8)
cons_ok = (cwx <= 8) & (cwy <= 8)
folded(nq_fwd, tight_ok, "forward, tight window <= 8x8")
folded(nq_fwd, cons_ok, "forward, record AABB in tile <= 8x8")
folded(nq_bwd, tight_ok, "backward, tight window <= 8x8")
folded(nq_bwd, cons_ok, "backward, record AABB in tile <= 8x8")
print(f"tight window: wx<=8 {int((wx <= 8).sum()) / n * 100:.1f}%  wy<=8 {int((wy <= 8).sum()) / n * 100:.1f}%  both {int(tight_ok.sum()) / n * 100:.1f}%;"
      f"  record AABB in tile: both {int(cons_ok.sum()) / n * 100:.1f}%")


# (a) larger tiles: unique (view, splat, coarse tile)
live = nq_bwd > 0
for kx, ky, nm in ((1, 1, "16x16"), (2, 1, "32x16"), (2, 2, "32x32")):
    key = ((vv * P + gid) * 4096 + (ty // ky)) * 4096 + (tx // kx)
    u = torch.unique(key).numel()
    ub = torch.unique(key[live]).numel()
    print(f"tiles {nm}: entries per view {u / NV:9.0f} ({u / n:.3f} of today);  entries the backward reduces {ub / NV:9.0f} "
          f"({ub / int(live.sum()):.3f} of today's {int(live.sum()) / NV:.0f})")
