#!/usr/bin/env python
"""Per-wave timeline of the latency-mapped (segmented) BACKWARD of ONE 1080p view (config 2) from the diagnostic build
tools/dbg/variants/r06_wave_timeline_bwd.patch (GGS_LIB_PATH=.../variants/wave_timeline_bwd.so): when every wave started and
ended (s_memtime, one origin per XCD), which (tile, quadrant, segment) it walked -- the block -> work mapping of render_bwd_body's
segmented form is replayed here from the bin header -- and how the kernel's length follows from that.  Gradients of this build are
not valid (the diagnostic words overwrite the SplatAux slots)."""
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
SEG, MAX_SEG = 64, 24
NQ = 1          # blocks per tile of the segmented backward: one tile wave per (tile, segment)
verts, faces = S.skirt_mesh(200, 250)
P = faces.shape[0]
model = MeshGaussianModel.from_tensors(verts, faces, S.skirt_gaussian_params(P, sh_degree=0), sh_degree=0, device=dev)
cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cams = S.stack_cameras(S.rig_cameras(n_rings=5, n_az=32, width=W, height=H, f=1500.0)[cam_i:cam_i + 1], device=dev)
bg = torch.zeros(3, device=dev)
w = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
with torch.no_grad():
    inp = dict(means3D=model.get_xyz, scales=model.get_scaling, rotations=model.get_rotation, opacities=model.get_opacity,
               shs=model.get_features)
for _ in range(3):
    color, radii, depth, alpha, st = R.forward_views(inp["means3D"], inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"],
                                                     None, view=cams["view"], proj=cams["proj"], campos=cams["campos"],
                                                     tanfov=cams["tanfov"], bg=bg, W=W, H=H, sh_degree=0)
    R.backward_views(st, w, want_means2D=False)
    torch.cuda.synchronize()
sec = R.bin_sections(st)
T = sec["tile_count"].shape[1]
order = sec["order"].cpu().numpy().astype(np.int64)[:T]
count = sec["tile_count"].cpu().numpy().reshape(-1)
bucket = st.bin[64:64 + 64].view(torch.int32).cpu().numpy().astype(np.int64)
nc = R.img_sections(st)["n_contrib"][0].cpu().numpy()
off = (P * 48 + 255) & ~255
dbg = st.geom[off:off + T * NQ * 16].view(torch.int32).cpu().numpy().astype(np.int64).reshape(T * NQ, 4) & 0xffffffff
t0, t1, hw, xcc = dbg[:, 0], dbg[:, 1], dbg[:, 2], dbg[:, 3]
# block -> (rank, segment): ggs_seg_item of csrc/ggs_common.h replayed
E = int(bucket[15]); n_ne = T - E; stride = ((E // n_ne) & ~1) + 1 if n_ne else 1
n_spare = T - n_ne
lenlo = [3072, 2048, 1536, 1024, 768, 512, 384, 256, 192, 128, 96, 64]
cum = np.cumsum(bucket[:12])
segmented = st.img.numel() > 2 * ((H * W * 4 + 255) & ~255)
blk = np.arange(T * NQ); rank = blk // NQ; q0 = blk % NQ
seg = np.zeros(T * NQ, np.int64); works = rank < n_ne
n_extra, offset, c = 0, 0, 11
if segmented:
    e = rank - n_ne
    for k in range(1, MAX_SEG):
        while c > 0 and lenlo[c - 1] <= k * SEG:
            c -= 1
        m = int(cum[c])
        if m == 0 or offset + m > n_spare:
            break
        n_extra = k
        xs = (rank >= n_ne) & (e >= offset) & (e < offset + m)
        seg[xs] = k
        rank = np.where(xs, e - offset, rank)
        works = works | xs
        offset += m
n_long = int(cum[11])
tile = order[np.minimum(rank, n_ne - 1) * stride]
L = count[tile]
gx = (W + 15) // 16
# last contributor of the wave's quadrant -> the positions it really walks
ncq = np.zeros(T * NQ, np.int64)
ty, tx = tile // gx, tile % gx
ncpad = np.zeros((((H + 15) // 16) * 16, gx * 16), np.int64); ncpad[:H, :W] = nc
for q in range(NQ):
    m = q0 == q
    y0, x0 = ty[m] * 16 + ((q // 2) * 8 if NQ == 4 else 0), tx[m] * 16 + ((q % 2) * 8 if NQ == 4 else 0)
    blkmax = np.zeros(m.sum(), np.int64)
    ext = 8 if NQ == 4 else 16
    for dy in range(ext):
        for dx in range(ext):
            blkmax = np.maximum(blkmax, ncpad[y0 + dy, x0 + dx])
    ncq[m] = blkmax
lo = seg * SEG
hi = np.where(seg == n_extra, 1 << 30, lo + SEG)
walked = np.clip(np.minimum(ncq, hi) - lo, 0, None) * works
TICKS_PER_US = float(os.environ.get("GGS_TICKS_PER_US", "2400"))
xcd_of = xcc & 15
start = np.zeros(len(t0))
for x in np.unique(xcd_of):
    m = xcd_of == x
    rel = ((t0[m] - t0[m][0] + (1 << 31)) & 0xffffffff) - (1 << 31)
    start[m] = (rel - rel.min()) / TICKS_PER_US
dur = ((t1 - t0) & 0xffffffff) / TICKS_PER_US
end = start + dur
ne = walked > 0
print(f"one 1080p view (camera {cam_i}): {T} tiles, {n_ne} non-empty, {int(count.sum())} list entries, {n_long} lists of >= {SEG}; up to 1 + {n_extra} segments per list; "
      f"{int(ne.sum())} waves with work of {T * NQ} ({'one wave per (tile, quadrant, segment)' if NQ == 4 else 'one wave per (tile, segment)'}, segments of {SEG})")
print(f"kernel span by the waves' own clocks: last start {start.max():.1f} us, last end {end.max():.1f} us (of the waves with work: {end[ne].max():.1f})")
print(f"waves with work: positions walked median {np.median(walked[ne]):.0f} / max {walked[ne].max()}; start median {np.median(start[ne]):.1f} / 90th "
      f"{np.percentile(start[ne], 90):.1f} / max {start[ne].max():.1f} us; duration median {np.median(dur[ne]):.1f} / 90th {np.percentile(dur[ne], 90):.1f} / max {dur[ne].max():.1f} us")
for s_ in range(n_extra + 1):
    m = ne & (seg == s_)
    if m.any():
        print(f"  segment {s_}: {int(m.sum())} waves, start median {np.median(start[m]):.1f} / max {start[m].max():.1f}, duration median {np.median(dur[m]):.1f} / max {dur[m].max():.1f}, "
              f"end max {end[m].max():.1f} us")
idx = np.argsort(-end * ne)[:10]
print("last waves to finish: " + ", ".join(f"seg {seg[i]} walked {walked[i]} start {start[i]:.1f} dur {dur[i]:.1f}" for i in idx))
idx = np.argsort(-dur * ne)[:10]
print("longest waves: " + ", ".join(f"seg {seg[i]} walked {walked[i]} L={L[i]} dur {dur[i]:.1f} (start {start[i]:.1f})" for i in idx))
lng = ne & (walked >= 60)
print(f"walks of >= 60 positions: {int(lng.sum())} waves, ns per position median {np.median(dur[lng] / walked[lng]) * 1e3:.0f}, fastest {np.min(dur[lng] / walked[lng]) * 1e3:.0f}, "
      f"slowest {np.max(dur[lng] / walked[lng]) * 1e3:.0f}")
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
unit = (((xcc & 15) * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
units, inv = np.unique(unit[ne], return_inverse=True)
load = np.bincount(inv, weights=walked[ne].astype(float)); nw = np.bincount(inv)
last = np.zeros(len(units)); np.maximum.at(last, inv, end[ne])
print(f"SIMDs that ran waves with work: {len(units)}; waves per SIMD mean {nw.mean():.2f} / max {nw.max()}; positions per SIMD mean {load.mean():.0f} / max {load.max():.0f}; "
      f"end of a SIMD's last wave: median {np.median(last):.1f} / 90th {np.percentile(last, 90):.1f} / max {last.max():.1f} us; corr(load, finish) {np.corrcoef(load, last)[0, 1]:.2f}")
grid = np.arange(0, min(float(end[ne].max()), 300.0), 2.0)
print("waves with work in flight every 2 us: " + " ".join(str(int(((start[ne] <= g) & (end[ne] > g)).sum())) for g in grid))
idle = ~ne
print(f"waves without work: {int(idle.sum())}, last start {start[idle].max():.1f} us, duration median {np.median(dur[idle]):.2f} us")
