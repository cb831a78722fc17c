"""CPU estimate of SIMD lane efficiency of the compositing passes on the config-2 scene (no occlusion):
for every visible splat, the pixels with alpha >= 1/255 inside its 3-sigma square, and how many 8x8 quadrants,
8x4 halves and 4x4 blocks those pixels touch."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import synthetic as S
from ggsplat.mesh_gaussian_model import MeshGaussianModel
from oracle import torch_oracle as TO, host_oracle as HO

v, f = S.skirt_mesh(); p = S.skirt_gaussian_params(f.shape[0], 0)
m = MeshGaussianModel.from_tensors(v, f, p, 0, device="cpu")
cams = S.rig_cameras()
W, H = 1920, 1080
for vi in (0, 100):
    ck = S.stack_cameras([cams[vi]], device="cpu")
    with torch.no_grad():
        xyz, sc, rot = HO.mesh_bind(m.mesh.v, m.mesh.f, m.binding, m._xyz, m._scaling, m._rotation)[:3]
        g = TO.preprocess(xyz, None, m.get_opacity, m.get_features, None, sc, rot, None,
                          ck["view"][0], ck["proj"][0], ck["campos"][0], W=W, H=H, tanfovx=float(ck["tanfov"][0, 0]),
                          tanfovy=float(ck["tanfov"][0, 1]), sh_degree=0)
    ok = g["valid"].numpy()
    px, py = g["px"].numpy()[ok], g["py"].numpy()[ok]
    con, op, rad = g["conic"].numpy()[ok], g["opacity"].numpy()[ok], g["radius"].numpy()[ok]
    R = int(rad.max())
    print(f"view {vi}: visible {ok.sum()}, radius mean {rad.mean():.1f} max {R}")
    tot_px = tot_q = tot_h = tot_b = tot_t = 0
    B = 2000
    for s in range(0, len(px), B):
        e = slice(s, s + B)
        cx, cy = np.round(px[e]).astype(int), np.round(py[e]).astype(int)
        o = np.arange(-R, R + 1)
        X = cx[:, None, None] + o[None, None, :]
        Y = cy[:, None, None] + o[None, :, None]
        dx, dy = px[e][:, None, None] - X, py[e][:, None, None] - Y
        pw = -0.5 * (con[e, 0][:, None, None] * dx * dx + con[e, 2][:, None, None] * dy * dy) - con[e, 1][:, None, None] * dx * dy
        al = np.minimum(0.99, op[e][:, None, None] * np.exp(pw))
        r = rad[e][:, None, None]
        hit = (pw <= 0) & (al >= 1 / 255) & (X >= 0) & (X < W) & (Y >= 0) & (Y < H) & (np.abs(dx) <= r + 1) & (np.abs(dy) <= r + 1)
        Xb, Yb = np.broadcast_to(X, hit.shape), np.broadcast_to(Y, hit.shape)
        n = np.nonzero(hit)
        gi, xs, ys = n[0], Xb[n], Yb[n]
        tot_px += len(gi)
        for name, sx, sy in (("q", 8, 8), ("h", 8, 4), ("b", 4, 4), ("t", 16, 16)):
            key = (gi.astype(np.int64) << 40) | ((xs // sx).astype(np.int64) << 20) | (ys // sy)
            c = len(np.unique(key))
            if name == "q": tot_q += c
            elif name == "h": tot_h += c
            elif name == "b": tot_b += c
            else: tot_t += c
    print(f"  contributing (splat, pixel) pairs {tot_px}; 16x16 tiles {tot_t}; 8x8 quadrants {tot_q} (lane eff {tot_px/(64*tot_q):.2%}); "
          f"8x4 halves {tot_h} (eff {tot_px/(32*tot_h):.2%}, halves/quadrant {tot_h/tot_q:.2f}); 4x4 blocks {tot_b} (eff {tot_px/(16*tot_b):.2%}, blocks/quadrant {tot_b/tot_q:.2f})")
