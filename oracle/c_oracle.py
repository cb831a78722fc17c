"""ctypes wrapper of oracle/libsplat_oracle.so (the C restatement, splat_oracle.c).

TEST INFRASTRUCTURE ONLY (see the header of splat_oracle.c): used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
Parity unpinned -- see splat_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsplat_oracle.so")
_lib = None


class GgoParams(C.Structure):
    _fields_ = [("P", C.c_int), ("K", C.c_int), ("deg", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "splat_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libsplat_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ggo_forward.restype = C.c_void_p
        _lib.ggo_num_rendered.restype = C.c_int64
        _lib.ggo_num_rendered.argtypes = [C.c_void_p]
        _lib.ggo_num_blended.restype = C.c_int64
        _lib.ggo_num_blended.argtypes = [C.c_void_p]
        _lib.ggo_free.argtypes = [C.c_void_p]
    return _lib


def _f(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class COracle:
    """One forward (kept as C state) + any number of backward calls."""

    def __init__(self, *, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
                 sh_degree=0, scale_modifier=1.0):
        L = lib()
        self.m3 = _f(means3D); self.op = _f(opacities).reshape(-1)
        self.shs = _f(shs); self.col = _f(colors_precomp)
        self.sc = _f(scales); self.rot = _f(rotations); self.cov = _f(cov3D_precomp)
        self.view = _f(viewmatrix).reshape(16); self.proj = _f(projmatrix).reshape(16)
        self.campos = _f(campos).reshape(3); self.bg = _f(bg).reshape(3)
        P = self.m3.shape[0]
        K = self.shs.shape[1] if self.shs is not None else 0
        assert (self.shs is None) != (self.col is None)
        assert (self.cov is None) != (self.sc is None or self.rot is None)
        self.prm = GgoParams(P, K, int(sh_degree), int(W), int(H), float(tanfovx), float(tanfovy), float(scale_modifier))
        self.P, self.K, self.W, self.H = P, K, int(W), int(H)
        self.color = np.zeros((3, H, W), np.float32)
        self.depth = np.zeros((1, H, W), np.float32)
        self.alpha = np.zeros((1, H, W), np.float32)
        self.radii = np.zeros(P, np.int32)
        self.h = C.c_void_p(L.ggo_forward(C.byref(self.prm), _p(self.bg), _p(self.m3), _p(self.shs), _p(self.col),
                                          _p(self.op), _p(self.sc), _p(self.rot), _p(self.cov), _p(self.view),
                                          _p(self.proj), _p(self.campos), _p(self.color), _p(self.depth),
                                          _p(self.alpha), _p(self.radii)))
        self.num_rendered = int(L.ggo_num_rendered(self.h))
        self.num_blended = int(L.ggo_num_blended(self.h))      # (Gaussian, pixel) pairs the compositing blended

    def internals(self) -> Dict[str, np.ndarray]:
        P, W, H = self.P, self.W, self.H
        T = ((W + 15) // 16) * ((H + 15) // 16)
        out = dict(xy=np.zeros((P, 2), np.float32), depth=np.zeros(P, np.float32),
                   conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
                   cov3d=np.zeros((P, 6), np.float32), tile_start=np.zeros(T + 1, np.int64),
                   list=np.zeros(max(self.num_rendered, 1), np.uint32), final_T=np.zeros((H, W), np.float32),
                   n_contrib=np.zeros((H, W), np.uint32))
        lib().ggo_get_internals(self.h, *[_p(out[k]) for k in ("xy", "depth", "conic_opacity", "rgb", "cov3d",
                                                               "tile_start", "list", "final_T", "n_contrib")])
        out["list"] = out["list"][:self.num_rendered]
        return out

    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None) -> Dict[str, np.ndarray]:
        P, K = self.P, self.K
        dc, dd, da = _f(dL_dcolor), _f(dL_ddepth), _f(dL_dalpha)
        g = dict(means2D=np.zeros((P, 3), np.float32), colors=np.zeros((P, 3), np.float32),
                 opacities=np.zeros((P, 1), np.float32), means3D=np.zeros((P, 3), np.float32),
                 cov3D=np.zeros((P, 6), np.float32), shs=np.zeros((P, max(K, 1), 3), np.float32),
                 scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
        lib().ggo_backward(self.h, _p(self.bg), _p(self.m3), _p(self.shs), _p(self.col), _p(self.sc), _p(self.rot),
                           _p(self.cov), _p(self.view), _p(self.proj), _p(self.campos), _p(dc), _p(dd), _p(da),
                           _p(g["means2D"]), _p(g["colors"]), _p(g["opacities"]), _p(g["means3D"]), _p(g["cov3D"]),
                           _p(g["shs"]) if K else None, _p(g["scales"]), _p(g["rotations"]))
        if not K:
            g["shs"] = None
        return g

    def close(self):
        if self.h:
            lib().ggo_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
