#!/usr/bin/env python
"""Per-wave timeline of the per-quadrant forward of ONE 1080p view (config 2) from the diagnostic build
tools/dbg/variants/r05_wave_timeline_fwd_quad.patch (GGS_LIB_PATH=.../variants/wave_timeline.so): when every wave started and
ended (s_memtime: shader-clock counts, one origin per XCD), on which XCD / CU / SIMD it ran, how long its list was.  Answers: is the kernel as long as its longest
walk, as long as its most loaded SIMD, or as long as its dispatch?  Forward only (the diagnostic words overwrite the SplatAux slots)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ggsplat import rasterizer as R, synthetic as S  # noqa: E402
from ggsplat.mesh_gaussian_model import MeshGaussianModel  # noqa: E402

dev = torch.device("cuda:0")
W, H = 1920, 1080
verts, faces = S.skirt_mesh(200, 250)
P = faces.shape[0]
model = MeshGaussianModel.from_tensors(verts, faces, S.skirt_gaussian_params(P, sh_degree=0), sh_degree=0, device=dev)
cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cams = S.stack_cameras(S.rig_cameras(n_rings=5, n_az=32, width=W, height=H, f=1500.0)[cam_i:cam_i + 1], device=dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    inp = dict(means3D=model.get_xyz, scales=model.get_scaling, rotations=model.get_rotation, opacities=model.get_opacity,
               shs=model.get_features)
for _ in range(3):
    color, radii, depth, alpha, st = R.forward_views(inp["means3D"], inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"],
                                                     None, view=cams["view"], proj=cams["proj"], campos=cams["campos"],
                                                     tanfov=cams["tanfov"], bg=bg, W=W, H=H, sh_degree=0)
    torch.cuda.synchronize()
sec = R.bin_sections(st)
T = sec["tile_count"].shape[1]
order = sec["order"].cpu().numpy().astype(np.int64)[:T]
count = sec["tile_count"].cpu().numpy().reshape(-1)
off = (P * 48 + 255) & ~255
dbg = st.geom[off:off + T * 4 * 16].view(torch.int32).cpu().numpy().astype(np.int64).reshape(T * 4, 4) & 0xffffffff
t0, t1, hw, xcc = dbg[:, 0], dbg[:, 1], dbg[:, 2], dbg[:, 3]
L = count[order][np.arange(T * 4) // 4]                      # list length of the tile each block worked on
# s_memtime on gfx950: one count per shader clock (calibrated below against the kernel's duration: ~2.4 counts per ns), and every
# XCD counts from its own origin (the raw starts of the eight XCDs are seconds apart): starts are taken relative to the earliest
# start ON THE SAME XCD.  32-bit words: differences modulo 2^32.
TICKS_PER_US = float(os.environ.get("GGS_TICKS_PER_US", "2400"))
xcd_of = dbg[:, 3] & 15
start = np.zeros(len(t0))
for x in np.unique(xcd_of):
    m = xcd_of == x
    rel = ((t0[m] - t0[m][0] + (1 << 31)) & 0xffffffff) - (1 << 31)
    start[m] = (rel - rel.min()) / TICKS_PER_US
dur = ((t1 - t0) & 0xffffffff) / TICKS_PER_US
end = start + dur
ne = L > 0
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
xcd = xcc & 15
unit = ((xcd * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
print(f"one 1080p view: {T} tiles, {int((count > 0).sum())} non-empty, {int(count.sum())} list entries; {T * 4} waves")
print(f"kernel span by the waves' own clocks: first start 0, last start {start.max():.1f} us, last end {end.max():.1f} us")
print(f"non-empty waves: start median {np.median(start[ne]):.1f} / max {start[ne].max():.1f} us; duration median {np.median(dur[ne]):.1f} / "
      f"90th {np.percentile(dur[ne], 90):.1f} / max {dur[ne].max():.1f} us; empty waves: last start {start[~ne].max():.1f} us, duration median {np.median(dur[~ne]):.2f} us")
idx = np.argsort(-dur)[:12]
print("longest waves: " + ", ".join(f"L={L[i]} {dur[i]:.1f}us (start {start[i]:.1f})" for i in idx))
# per-entry cost of a walk: duration / list length for the long lists
lng = ne & (L >= 400)
print(f"lists >= 400 entries: {int(lng.sum())} waves, us per list entry median {np.median(dur[lng] / L[lng]) * 1e3:.0f} ns, "
      f"fastest {np.min(dur[lng] / L[lng]) * 1e3:.0f}, slowest {np.max(dur[lng] / L[lng]) * 1e3:.0f}")
# load per SIMD: sum of the non-empty waves' list lengths, and when the SIMD's last wave ended
units, inv = np.unique(unit[ne], return_inverse=True)
load = np.bincount(inv, weights=L[ne].astype(float))
nw = np.bincount(inv)
last = np.zeros(len(units)); np.maximum.at(last, inv, end[ne])
print(f"SIMDs that ran non-empty waves: {len(units)}; waves per SIMD mean {nw.mean():.2f} / max {nw.max()}; list entries per SIMD mean "
      f"{load.mean():.0f} / max {load.max():.0f}; end of a SIMD's last wave: median {np.median(last):.1f} / 90th {np.percentile(last, 90):.1f} / max {last.max():.1f} us")
c = np.corrcoef(load, last)[0, 1]
print(f"correlation(load of a SIMD, its finish time) = {c:.2f}")
worst = np.argsort(-last)[:6]
for w in worst:
    members = np.where(ne & (unit == units[w]))[0]
    print(f"  SIMD {units[w]}: finishes {last[w]:.1f} us, {len(members)} waves, lists " + " ".join(str(L[m]) for m in members) + " | durations " + " ".join(f"{dur[m]:.0f}" for m in members))
# how many waves are alive over time
grid = np.arange(0, min(float(end[ne].max()), 400.0), 2.0)
alive = [(int(((start[ne] <= g) & (end[ne] > g)).sum())) for g in grid]
print("non-empty waves in flight every 2 us: " + " ".join(str(a) for a in alive))
