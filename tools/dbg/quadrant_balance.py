"""Census for a splat-major backward (GPU): per (tile, quadrant) number of list entries the forward kept in the
quadrant's narrowed mask, and what a row-per-quadrant systolic walk would cost in steps:
  ideal    = sum over tiles of sum_q c_q / 4          (perfectly balanced rows, no fill)
  per tile = sum over tiles of max_q c_q + 15         (row r = quadrant r of one tile, 15 fill steps)
"""
import sys, os, math, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S, rasterizer as R
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev = "cuda"; W, H = 1920, 1080
v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device=dev)
cams = S.rig_cameras()
NV = 8
sel = [0, 23, 47, 64, 90, 111, 130, 159]
ck = S.stack_cameras([cams[i] for i in sel], device=dev)
with torch.no_grad():
    m.update_face_coor()
    color, radii, depth, alpha, st = R.forward_views(m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None,
        view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
sec = R.bin_sections(st)
n = st.num_rendered
cnt = sec["tile_count"].reshape(-1).cpu().numpy().astype(np.int64)          # [V*T]
words = (sec["ids"][:n].to(torch.int64) & 0xffffffff).cpu().numpy()
assert cnt.sum() == n, (cnt.sum(), n)
tile_of = np.repeat(np.arange(len(cnt)), cnt)          # entries are laid out view-major, tile after tile?  check with offsets
off = sec["tile_offset"].reshape(NV, -1).cpu().numpy().astype(np.int64)
vb = sec["view_base"].cpu().numpy().astype(np.int64)
T = off.shape[1]
start = (vb[:, None] + off).reshape(-1)
order = np.argsort(start, kind="stable")
# entry index -> tile via searchsorted over the starts of non-empty tiles
ne = cnt > 0
st_ne = start[ne]; id_ne = np.nonzero(ne)[0]
o = np.argsort(st_ne); st_ne = st_ne[o]; id_ne = id_ne[o]
tile_of = id_ne[np.searchsorted(st_ne, np.arange(n), side="right") - 1]
mask = (words >> 28) & 15
cq = np.zeros((len(cnt), 4), np.int64)
for q in range(4):
    np.add.at(cq[:, q], tile_of, (mask >> q) & 1)
tot = cq.sum()
act = cq.sum(1) > 0
ideal = tot / 4
per_tile = (cq[act].max(1) + 15).sum()
per_tile_nofill = cq[act].max(1).sum()
print(f"views {NV}: entries {n}, entries with mask != 0 {int((mask != 0).sum())}, quadrant passes {tot} = {tot / n:.2f} per entry")
print(f"non-empty tiles {int(act.sum())}, mean c_q {cq[act].mean():.1f}, mean max_q {cq[act].max(1).mean():.1f}")
print(f"steps: ideal {ideal:.0f}; row = quadrant of one tile: {per_tile} (x{per_tile / ideal:.2f}), without the fill {per_tile_nofill} (x{per_tile_nofill / ideal:.2f})")
# two tiles per wave, rows paired greedily: rows take (tile A q, tile B q') chains: cost = max over rows of the chained length + 15
# simple variant: a wave takes one tile, rows chain quadrants of the NEXT tile in LPT order when they run dry: bound by a
# queue model: 4 rows pull (tile, quadrant) items from a per-wave queue of G tiles
for G in (2, 4, 8):
    idx = np.nonzero(act)[0]
    srt = idx[np.argsort(-cq[idx].sum(1), kind="stable")]
    steps = 0
    for g0 in range(0, len(srt), G):
        items = np.sort(cq[srt[g0:g0 + G]].reshape(-1))[::-1]
        rows = np.zeros(4, np.int64)
        for c in items:
            if c: rows[np.argmin(rows)] += c
        steps += rows.max() + 15
    print(f"  {G} tiles per wave, quadrant items LPT over 4 rows, chained: {steps} (x{steps / ideal:.2f})")
