"""Tile-list statistics of the config-2 workload (GPU): list lengths, processed lengths."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras()
for vi in (0, 40, 100, 159):
    ck = S.stack_cameras([cams[vi]], device=dev)
    with torch.no_grad():
        m.update_face_coor()
        color, radii, depth, alpha, st = R.forward_views(m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None,
            view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=1920, H=1080, sh_degree=0)
    sec = R.bin_sections(st)
    cnt = sec["tile_count"][0].cpu().numpy().astype(np.int64)
    HW = 1920 * 1080
    ncon = R.img_sections(st)["n_contrib"][0]
    nc = torch.nn.functional.pad(ncon, (0, 0, 0, 8))  # H 1080 -> 1088
    tmax = nc.reshape(68, 16, 120, 16).permute(0, 2, 1, 3).reshape(68 * 120, 256).max(1).values.cpu().numpy()
    act = cnt > 0
    pct = lambda a: [int(np.percentile(a, q)) for q in (50, 90, 99, 100)]
    print(f"view {vi}: N={cnt.sum()} active tiles={act.sum()} L pct50/90/99/max={pct(cnt[act])} processed(max n_contrib per tile) pct={pct(tmax[act])} "
          f"sum L={cnt.sum()} sum processed={tmax.sum()} mean n_contrib/covered px={float(ncon[ncon>0].float().mean()):.1f} covered px={int((ncon>0).sum())} radii mean={float(radii.float().mean()):.1f}")

# ---- quadrant-mask statistics for the last view ----
sec = R.bin_sections(st)
cnt = sec["tile_count"][0].cpu().numpy().astype(np.int64)
off = sec["tile_offset"][0].cpu().numpy().astype(np.int64)
words = sec["ids"].cpu().numpy().astype(np.uint32)[:int(cnt.sum())]
recs = st.geom[:100000 * 48].view(torch.int32).reshape(100000, 12).cpu().numpy()
tile_of = np.repeat(np.arange(len(cnt)), cnt)
# list position within tile
posn = np.arange(len(words)) - np.repeat(off, cnt)
processed = posn < np.repeat(tmax, cnt)
gid = words & 0x0fffffff
blend = (words >> 28).astype(np.int64)
bbx, bby = recs[gid, 10].astype(np.int64), recs[gid, 11].astype(np.int64)
sx = lambda w: ((w & 0xffff) ^ 0x8000) - 0x8000
xmin, xmax, ymin, ymax = sx(bbx), bbx >> 16, sx(bby), bby >> 16
ox, oy = (tile_of % 120) * 16, (tile_of // 120) * 16
hx0 = (xmin <= ox + 7) & (xmax >= ox); hx1 = (xmin <= ox + 15) & (xmax >= ox + 8)
hy0 = (ymin <= oy + 7) & (ymax >= oy); hy1 = (ymin <= oy + 15) & (ymax >= oy + 8)
aabb_bits = (hx0 & hy0).astype(int) + (hx1 & hy0) + (hx0 & hy1) + (hx1 & hy1)
pop = np.array([bin(i).count("1") for i in range(16)])
bl_bits = pop[blend]
p = processed
print(f"entries={len(words)} processed={p.sum()} AABB quadrants/processed entry={aabb_bits[p].mean():.2f} "
      f"blended quadrants/processed entry={bl_bits[p].mean():.2f} entries with no blend={(bl_bits[p]==0).mean():.2%} "
      f"blended quadrants / AABB quadrants={bl_bits[p].sum()/aabb_bits[p].sum():.2%}")
