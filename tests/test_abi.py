"""C-ABI library: loads without a GPU, exports every symbol include/ggsplat.h declares, and its
host-only entry points (size queries, argument validation) behave.  No compute is launched here."""
import ctypes as C
import os
import re

import pytest

from ggsplat import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ggsplat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ggs_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert b"gfx950" in L.ggs_version()


def test_workspace_sizes_and_layout():
    L = _lib.lib()
    prm = _lib.GgsParams(100000, 1, 0, 1920, 1080, 4, 1.0, 0, 0)
    g, i, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert L.ggs_workspace_sizes(C.byref(prm), 1 << 20, C.byref(g), C.byref(i), C.byref(b)) == 0
    assert g.value >= 4 * 100000 * 48 and i.value >= 2 * 4 * 1920 * 1080 * 4
    off = (C.c_size_t * 8)()
    assert L.ggs_bin_layout(C.byref(prm), 1 << 20, off) == 0
    o = list(off)
    assert o == sorted(o) and o[0] == 0 and o[7] == b.value
    T = 120 * 68
    assert o[2] - o[1] >= 4 * T * 4 and o[6] - o[5] >= (1 << 20) * 8 and o[7] - o[6] >= (1 << 20) * 4
    assert all(x % 256 == 0 for x in o)
    assert L.ggs_backward_scratch_bytes(C.byref(prm)) >= 4 * 100000 * 48


def test_argument_validation_happens_before_any_launch():
    L = _lib.lib()
    bad = _lib.GgsParams(10, 1, 0, 0, 64, 1, 1.0, 0, 0)            # W = 0
    assert L.ggs_workspace_sizes(C.byref(bad), 16, None, None, None) == -1
    assert b"bad sizes" in L.ggs_last_error()
    prm = _lib.GgsParams(10, 1, 0, 64, 64, 1, 1.0, 0, 0)
    one = C.c_void_p(16)                                           # never dereferenced: validation fails first
    # both shs and colors given
    args = [one] * 14
    rc = L.ggs_forward(C.byref(prm), *args, 16, *([one] * 6))
    assert rc == -1 and b"SHs or precomputed colors" in L.ggs_last_error()
    # neither scales/rots nor cov
    a = [one, one, one, None, one, None, None, None, one, one, one, one, one, one]
    rc = L.ggs_forward(C.byref(prm), *a, 16, *([one] * 6))
    assert rc == -1 and b"scale/rotation pair or precomputed 3D covariance" in L.ggs_last_error()
    # sh_degree needs more coefficients than K
    prm2 = _lib.GgsParams(10, 4, 3, 64, 64, 1, 1.0, 0, 0)
    a = [one, one, one, None, one, one, one, None, one, one, one, one, one, one]
    rc = L.ggs_forward(C.byref(prm2), *a, 16, *([one] * 6))
    assert rc == -1 and b"needs 16 coefficients" in L.ggs_last_error()
    assert L.ggs_mesh_bind_forward(-1, 0, *([None] * 11)) == -1


def test_product_refuses_cpu_tensors():
    """No CPU fallback in the product: CPU tensors raise instead of silently computing."""
    import torch
    from ggsplat import rasterizer as R
    with pytest.raises(_lib.GgsError, match="GPU"):
        R.forward_views(torch.zeros(4, 3), torch.zeros(4, 1), torch.zeros(4, 1, 3), None, torch.ones(4, 3),
                        torch.zeros(4, 4), None, view=torch.eye(4).reshape(1, 16), proj=torch.eye(4).reshape(1, 16),
                        campos=torch.zeros(1, 3), tanfov=torch.ones(1, 2), bg=torch.zeros(3), W=32, H=32, sh_degree=0)
    from ggsplat.mesh_gaussian_model import mesh_bind
    with pytest.raises(RuntimeError, match="GPU only"):
        mesh_bind(torch.zeros(3, 3), torch.zeros(1, 3, dtype=torch.long), torch.zeros(1, dtype=torch.long),
                  torch.zeros(1, 3), torch.zeros(1, 3), torch.ones(1, 4))


def test_product_does_not_import_the_oracle():
    """Nothing under gaussian-garments_amd/ may import, load or execute oracle/ (checker only)."""
    pkg = os.path.join(ROOT, "gaussian-garments_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")) or "CDLL(" in s or "dlopen(" in s:
                        assert "oracle" not in s, f"{f}: {s}"
