"""ctypes binding of libggsplat.so (include/ggsplat.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load,
every entry point raises.  PyTorch is imported first so that the library binds to the
HIP runtime PyTorch already loaded (same SONAME libamdhip64.so.7) -- device pointers,
streams and the caching allocator are PyTorch's; the kernels are ours.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below: one HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GGS_LIB_PATH: load another build of the same library (A/B timing of kernel variants, tools/dbg/job.sh ab)
LIB_PATH = os.environ.get("GGS_LIB_PATH") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "libggsplat.so"))


class GgsParams(C.Structure):
    _fields_ = [("P", C.c_int), ("K", C.c_int), ("sh_degree", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("n_views", C.c_int), ("scale_modifier", C.c_float), ("prefiltered", C.c_int), ("debug", C.c_int)]


class GgsStepPrologue(C.Structure):
    """include/ggsplat.h GgsStepPrologue (ggs_step_prologue)."""
    _fields_ = [("n_clear", C.c_int), ("clear_ptr", C.c_void_p * 8), ("clear_bytes", C.c_size_t * 8),
                ("copy_src", C.c_void_p), ("copy_dst", C.c_void_p), ("copy_bytes", C.c_size_t),
                ("P", C.c_int), ("F", C.c_int), ("verts", C.c_void_p), ("faces", C.c_void_p), ("binding", C.c_void_p),
                ("local_xyz", C.c_void_p), ("log_scaling", C.c_void_p), ("raw_rot", C.c_void_p), ("bary", C.c_void_p),
                ("xyz", C.c_void_p), ("scaling", C.c_void_p), ("rotation", C.c_void_p),
                ("n_opacity", C.c_int), ("opacity_logit", C.c_void_p), ("opacity", C.c_void_p), ("consumer_mask", C.c_uint)]


class GgsStepTail(C.Structure):
    """include/ggsplat.h GgsStepTail (ggs_registration_aux_tail)."""
    _fields_ = [("loss_sums", C.c_void_p), ("header", C.c_void_p), ("out_block", C.c_void_p),
                ("n_adam_states", C.c_int), ("adam_states", C.c_void_p * 16), ("beta1", C.c_double), ("beta2", C.c_double)]


class GgsError(RuntimeError):
    pass


_lib = None

_PTR = C.c_void_p
_SIGS = {
    "ggs_workspace_sizes": (C.c_int, [C.POINTER(GgsParams), C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ggs_bin_layout": (C.c_int, [C.POINTER(GgsParams), C.c_size_t, C.POINTER(C.c_size_t)]),
    "ggs_backward_scratch_bytes": (C.c_size_t, [C.POINTER(GgsParams)]),
    "ggs_forward": (C.c_int, [C.POINTER(GgsParams)] + [_PTR] * 14 + [C.c_size_t] + [_PTR] * 6),
    "ggs_forward_count": (C.c_int, [C.POINTER(GgsParams)] + [_PTR] * 14 + [C.c_size_t] + [_PTR] * 6),
    "ggs_forward_render": (C.c_int, [C.POINTER(GgsParams)] + [_PTR] * 14 + [C.c_size_t] + [_PTR] * 6),
    "ggs_forward_stages": (C.c_int, [C.c_int, C.POINTER(GgsParams)] + [_PTR] * 14 + [C.c_size_t] + [_PTR] * 6),
    "ggs_forward_spec": (C.c_int, [C.POINTER(GgsParams)] + [_PTR] * 14 + [C.c_size_t] + [_PTR] * 8),
    "ggs_backward": (C.c_int, [C.POINTER(GgsParams)] + [_PTR] * 13 + [C.c_size_t] + [_PTR] * 13 + [C.c_int, _PTR]),
    "ggs_mesh_bind_forward": (C.c_int, [C.c_int, C.c_int] + [_PTR] * 11),
    "ggs_mesh_bind_backward": (C.c_int, [C.c_int, C.c_int] + [_PTR] * 15),
    "ggs_photometric_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ggs_photometric_forward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 6),
    "ggs_photometric_backward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 7),
    "ggs_photometric_forward_tab": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 6),
    "ggs_photometric_backward_tab": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 7),
    "ggs_photometric_forward_roi": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 9),
    "ggs_photometric_backward_roi": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 10),
    "ggs_photometric_forward_sparse": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 11),
    "ggs_mask_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 3),
    "ggs_dist2_3nn": (C.c_int, [C.c_int, _PTR, _PTR, _PTR]),
    "ggs_dist2_3nn_scratch_bytes": (C.c_size_t, [C.c_int]),
    "ggs_dist2_3nn_grid": (C.c_int, [C.c_int, _PTR, _PTR, _PTR, _PTR]),
    "ggs_fused_bias_act": (C.c_int, [C.c_size_t, _PTR, _PTR, _PTR, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _PTR, _PTR]),
    "ggs_fused_bias_act_t": (C.c_int, [C.c_int, C.c_size_t, _PTR, _PTR, _PTR, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _PTR, _PTR]),
    "ggs_upfirdn2d_t": (C.c_int, [C.c_int] * 5 + [_PTR, _PTR] + [C.c_int] * 10 + [_PTR, _PTR]),
    "ggs_upfirdn2d_out_size": (C.c_int, [C.c_int] * 12 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ggs_upfirdn2d": (C.c_int, [C.c_int] * 4 + [_PTR, _PTR] + [C.c_int] * 10 + [_PTR, _PTR]),
    "ggs_adam_state_bytes": (C.c_size_t, []),
    "ggs_adam_tick": (C.c_int, [_PTR, C.c_double, C.c_double, _PTR, _PTR]),
    "ggs_adam_step": (C.c_int, [C.c_size_t, _PTR, _PTR, _PTR, _PTR, _PTR, C.c_double, C.c_double, C.c_double, _PTR, _PTR, _PTR]),
    "ggs_adam_tick_multi": (C.c_int, [C.c_int, _PTR, C.c_double, C.c_double, _PTR, _PTR]),
    "ggs_adam_step_multi": (C.c_int, [C.c_int] + [_PTR] * 7 + [C.c_double] * 3 + [_PTR] * 2),
    "ggs_adam_tick_step_multi": (C.c_int, [C.c_int] + [_PTR] * 7 + [C.c_double] * 3 + [_PTR] * 2),
    "ggs_registration_aux": (C.c_int, [C.c_int] + [_PTR] * 7 + [C.c_float] * 4 + [_PTR] * 9),
    "ggs_registration_aux_tail": (C.c_int, [C.c_int] + [_PTR] * 7 + [C.c_float] * 4 + [_PTR] * 8 + [C.POINTER(GgsStepTail), _PTR]),
    "ggs_step_prologue": (C.c_int, [C.POINTER(GgsStepPrologue), _PTR]),
    "ggs_step_end": (C.c_int, []),
    "ggs_step_clear_plan": (C.c_int, [C.POINTER(GgsParams), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ggs_host_mapped_pointer": (C.c_int, [_PTR, C.POINTER(C.c_void_p)]),
    "ggs_visibility_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_size_t]),
    "ggs_visibility": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_PTR] * 6 + [C.c_size_t, _PTR, _PTR, _PTR]),
    "ggs_count_blends": (C.c_int, [C.POINTER(GgsParams), _PTR, _PTR, C.c_size_t, _PTR, _PTR, _PTR]),
    "ggs_count_pairs": (C.c_int, [C.POINTER(GgsParams), _PTR, _PTR, C.c_size_t, _PTR, _PTR, _PTR]),
    "ggs_count_forward_visits": (C.c_int, [C.POINTER(GgsParams), _PTR, _PTR, C.c_size_t, _PTR, _PTR]),
    "ggs_profile_enable": (C.c_int, [C.c_int]),
    "ggs_profile_read": (C.c_int, [C.POINTER(C.c_float), C.c_int]),
    "ggs_profile_stamps": (C.c_int, [_PTR, C.c_int]),
    "ggs_profile_stamp_log": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "ggs_last_error": (C.c_char_p, []),
    "ggs_version": (C.c_char_p, []),
    "ggs_build_id": (C.c_char_p, []),
    "ggs_tile_size": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load (once) and return the CDLL.  Raises GgsError if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GgsError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C gaussian-garments_amd/csrc`. There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise GgsError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ggs_last_error().decode("utf-8", "replace")
        raise GgsError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "ggsplat: tensor must be contiguous"
    return t.data_ptr()             # ctypes converts the int for the void* parameters (no c_void_p object per argument)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(dev) -> int:
    """HIP stream handle of PyTorch's current stream on `dev`, as an int (ctypes converts it for the void* parameters).
    torch.cuda.current_stream() builds a Python Stream object per call (~5 us, six calls per eager s2 step)."""
    if _raw_stream is not None:
        i = dev.index
        return _raw_stream(torch.cuda.current_device() if i is None else i)
    return torch.cuda.current_stream(dev).cuda_stream


def host_mapped_pointer(t) -> int:
    """Device address of a pinned host tensor (0 when the memory is not mapped into the device's address space)."""
    out = C.c_void_p()
    if not t.is_pinned() or lib().ggs_host_mapped_pointer(t.data_ptr(), C.byref(out)) != 0:
        return 0
    return int(out.value or 0)


_SRC_ORDER = ("ggs_pergauss.hip", "ggs_binning.hip", "ggs_render.hip", "ggs_mesh.hip", "ggs_loss.hip", "ggs_knn.hip",
              "ggs_stylegan.hip", "ggs_visibility.hip", "ggs_adam.hip", "ggs_regaux.hip", "ggs_api.hip", "ggs_common.h",
              "ggs_kernels.h", "ggs_render_common.h", os.path.join("..", "..", "include", "ggsplat.h"), "Makefile")


def source_hash() -> str:
    """The id `make` would bake into a library built from the sources on disk now (csrc/Makefile SRC_HASH)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
    for name in _SRC_ORDER:
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_id() -> str:
    return lib().ggs_build_id().decode()


_tile = None


def tile_size():
    """(width, height) in pixels of the tiles the loaded library bins into: (16, 16) in the product."""
    global _tile
    if _tile is None:
        w, h = C.c_int(), C.c_int()
        lib().ggs_tile_size(C.byref(w), C.byref(h))
        _tile = (w.value, h.value)
    return _tile


def n_tiles(W: int, H: int) -> int:
    tw, th = tile_size()
    return ((W + tw - 1) // tw) * ((H + th - 1) // th)


def dtype_code(dt) -> int:
    """GGS_DTYPE_* of a torch dtype for the typed StyleGAN-op entry points; anything but float / half / double raises
    like the upstream dispatch macro does."""
    import torch
    try:
        return {torch.float32: 0, torch.float16: 1, torch.float64: 2}[dt]
    except KeyError:
        raise RuntimeError(f'"ggsplat stylegan op" not implemented for \'{str(dt).replace("torch.", "")}\'') from None
