"""Host side of the rasterizer: workspace management + the autograd Function.

Mirrors the Python half of the reference's external extension
(`diff_gaussian_rasterization_depth_alpha`): an autograd Function whose forward
calls the native forward and keeps the opaque geom / binning / image buffers for
the native backward.  Call site being served: gaussian_renderer/__init__.py:54,
:103-111.  Differences, all MI355X-motivated:
  * the native side is a C ABI over caller-allocated workspaces (no resize callback);
  * one call can batch V views (grid dimension = view) over the same Gaussians;
  * the only host sync is one 16-byte header read-back per call after the tile scan
    (num_rendered + overflow flag), never per kernel; the compositing is queued behind it;
  * while the current stream is being captured into a hipGraph (torch.cuda.graph) there is NO sync: the call
    uses the binning capacity learnt by earlier eager calls, and the overflow flag stays on the device
    (`last_header()`), where the guarded optimiser kernels (ggsplat.adam) and the graph owner read it.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import GgsParams, check, lib, ptr

# running estimate of the binning capacity per (device, P, W, H, V); grows on overflow
_cap_hint: Dict[Tuple, int] = {}


def _header_of(binb: torch.Tensor) -> torch.Tensor:
    return binb[:16].view(torch.int64)


def last_header() -> Optional[torch.Tensor]:
    """Device int64[2] = GgsBinHeader {num_rendered, overflow} of the most recent forward ISSUED BY THE CALLING THREAD
    (aliases its workspace).  Inside a dL_dcolor_fn callback of ggsplat.batch.fwd_bwd_views it is the forward whose images the
    callback was handed, in every pipeline mode (the staged form, pipeline=2, has the NEXT set's stages queued by then:
    StagedForward.make_current).  Per thread, like the pinned landing buffer: the guarded optimiser step of one thread must
    not read the overflow word of a forward that another thread (an eval worker) issued in between."""
    b = getattr(_pinned, "last_bin", None)
    return None if b is None else _header_of(b[0])


def last_tile_count() -> Optional[torch.Tensor]:
    """List lengths of the 16x16 tiles of this thread's most recent forward: int32 [V, ceil(H/16) * ceil(W/16)], a view into
    that forward's binning buffer (which it keeps alive).  The region-of-interest form of the fused loss takes it
    (loss.fused_photometric_loss(..., tile_count=...)): dL/dimage is only needed where a tile has a list."""
    b = getattr(_pinned, "last_bin", None)
    if b is None or _lib.tile_size() != (16, 16):          # the region-of-interest loss is cut for 16 x 16 tiles
        return None
    binb, V, T, off = b
    return binb[off:off + V * T * 4].view(torch.int32).reshape(V, T)


# headers of the forwards issued while a stream was being captured (the graph owner checks their overflow words)
_capture_headers: list = []


def pop_capture_headers() -> list:
    """Device int64[2] headers {num_rendered, overflow} of every forward captured since the last call."""
    out = list(_capture_headers)
    _capture_headers.clear()
    return out


def grow_capacity(factor: float = 2.0) -> None:
    """Enlarge every learnt binning capacity (a captured step reported overflow: re-capture after this)."""
    for k in list(_cap_hint):
        _cap_hint[k] = int(_cap_hint[k] * factor) + 1024


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None or (t.dtype is torch.float32 and t.is_contiguous()):      # the common case costs two attribute reads
        return t
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_ws_sizes: Dict[Tuple, Tuple[int, int, int]] = {}
_bwd_scratch: Dict[Tuple, int] = {}          # (P, K, V) -> ggs_backward_scratch_bytes (one C call per shape, not per backward)
_tc_off: Dict[Tuple, int] = {}


def _tile_count_offset(L, prm, cap: int) -> int:
    """Byte offset of the tile_count section in the binning buffer (ggs_bin_layout section 1), memoised per problem shape."""
    k = (prm.P, prm.W, prm.H, prm.n_views, cap)
    r = _tc_off.get(k)
    if r is None:
        off = (C.c_size_t * 8)()
        check(L.ggs_bin_layout(C.byref(prm), cap, off), "ggs_bin_layout")
        r = _tc_off[k] = int(off[1])
    return r


def _workspace_sizes(L, prm, cap: int) -> Tuple[int, int, int]:
    """ggs_workspace_sizes, memoised per problem shape (the per-view loop asks the same question every call)."""
    k = (prm.P, prm.K, prm.W, prm.H, prm.n_views, cap)
    r = _ws_sizes.get(k)
    if r is None:
        gsz, isz, bsz = C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(L.ggs_workspace_sizes(C.byref(prm), cap, C.byref(gsz), C.byref(isz), C.byref(bsz)), "ggs_workspace_sizes")
        if len(_ws_sizes) > 256:
            _ws_sizes.clear()
        r = _ws_sizes[k] = (gsz.value, isz.value, bsz.value)
    return r


_pinned = threading.local()
_tanfov_cache: Dict = {}


def _pinned_header(dev) -> torch.Tensor:
    """Pinned int64[2] landing buffer of the header read-back, one per (thread, device): two threads (eval workers,
    multi-stream batching) calling forward_views concurrently must not read each other's num_rendered / overflow."""
    d = getattr(_pinned, "buf", None)
    if d is None:
        d = _pinned.buf = {}
    t = d.get(dev.index)
    if t is None:
        t = d[dev.index] = torch.empty(2, dtype=torch.int64, pin_memory=True)
    return t


def _header_event(dev) -> "torch.cuda.Event":
    """Event marking the header copy of a forward, one per (thread, device)."""
    d = getattr(_pinned, "ev", None)
    if d is None:
        d = _pinned.ev = {}
    e = d.get(dev.index)
    if e is None:
        e = d[dev.index] = torch.cuda.Event()
        e.record(torch.cuda.current_stream(dev))       # torch creates the HIP event lazily: the C side needs the handle
    return e


def tanfov_tensor(tanfovx: float, tanfovy: float, dev) -> torch.Tensor:
    """[1,2] device tensor of the two tangents, cached per camera intrinsics (avoids an H2D copy per call)."""
    k = (float(tanfovx), float(tanfovy), dev.index)
    t = _tanfov_cache.get(k)
    if t is None:
        if len(_tanfov_cache) > 4096:
            _tanfov_cache.clear()
        t = _tanfov_cache[k] = torch.tensor([[tanfovx, tanfovy]], dtype=torch.float32, device=dev)
    return t


def _stream_ptr(dev) -> int:
    return _lib.stream_ptr(dev)


class StepWorkspaces:
    """Workspaces of ONE forward + backward allocated up front (plan_step), so that a caller can name them -- and the two
    ranges the library zero-fills first -- to ggs_step_prologue before the calls run."""
    __slots__ = ("key", "cap", "geom", "img", "bin", "scratch", "bin_clear", "scratch_clear")


def plan_step(P: int, K: int, sh_degree: int, W: int, H: int, V: int, dev) -> StepWorkspaces:
    L = lib()
    ws = StepWorkspaces()
    ws.key = (dev.index, P, W, H, V)
    ws.cap = cap = _cap_hint.get(ws.key, max(8 * P * V, 1 << 16))
    prm = GgsParams(P, K, int(sh_degree), int(W), int(H), V, 1.0, 0, 0)
    gsz, isz, bsz = _workspace_sizes(L, prm, cap)
    new = (lambda n: torch.empty(n, device=dev, dtype=torch.uint8))
    ws.geom, ws.img, ws.bin = new(gsz), new(isz), new(bsz)
    ws.scratch = new(int(L.ggs_backward_scratch_bytes(C.byref(prm))))
    a, b = C.c_size_t(), C.c_size_t()
    check(L.ggs_step_clear_plan(C.byref(prm), cap, C.byref(a), C.byref(b)), "ggs_step_clear_plan")
    ws.bin_clear, ws.scratch_clear = int(a.value), int(b.value)
    return ws


class ForwardState:
    """Everything the backward needs (the reference keeps the same things in ctx)."""
    __slots__ = ("prm", "bg", "means3D", "shs", "colors", "opac", "scales", "rots", "cov", "view", "proj",
                 "campos", "tanfov", "geom", "bin", "cap", "img", "num_rendered", "bwd_args")

    @property
    def header(self) -> torch.Tensor:
        """Device int64[2] {num_rendered, overflow} of THIS forward (a view into its binning buffer, made on demand: the
        per-view loop never looks at it)."""
        return _header_of(self.bin)


_prm_cache: Dict[Tuple, GgsParams] = {}


def _params(P: int, K: int, deg: int, W: int, H: int, V: int, scale_modifier: float, debug: int) -> GgsParams:
    """The GgsParams block of a problem shape, built once (nothing writes to it after construction; a ctypes Structure
    with nine fields costs ~3 us per call to fill)."""
    k = (P, K, deg, W, H, V, scale_modifier, debug)
    prm = _prm_cache.get(k)
    if prm is None:
        if len(_prm_cache) > 256:
            _prm_cache.clear()
        prm = _prm_cache[k] = GgsParams(P, K, deg, W, H, V, scale_modifier, 0, debug)
    return prm


def _prep_forward(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, view, proj, campos, tanfov, bg,
                  W: int, H: int, sh_degree: int, scale_modifier: float, debug: bool):
    """Input normalisation shared by forward_views and StagedForward: float32 / contiguous tensors on the GPU, the upstream
    empty-tensor convention, stacked camera blocks, GgsParams and the four output tensors."""
    dev = means3D.device
    if dev.type != "cuda":
        raise _lib.GgsError("ggsplat: tensors must live on the GPU (there is no CPU path in the product)")
    means3D, opacities, shs, colors_precomp = _f32c(means3D), _f32c(opacities), _f32c(shs), _f32c(colors_precomp)
    scales, rotations, cov3D_precomp = _f32c(scales), _f32c(rotations), _f32c(cov3D_precomp)
    P = means3D.shape[0]
    if P > 0:                                   # upstream convention: a missing input may arrive as an empty tensor
        shs, colors_precomp, scales, rotations, cov3D_precomp = (
            None if (t is not None and t.numel() == 0) else t
            for t in (shs, colors_precomp, scales, rotations, cov3D_precomp))
    # camera tensors and background left on the host are moved (the kernels would dereference host pointers otherwise: a GPU
    # memory fault, not an exception); the reference keeps them on the device (scene/cameras.py ends every matrix in .cuda())
    di = dev.index
    if not (view.get_device() == di and proj.get_device() == di and campos.get_device() == di and tanfov.get_device() == di
            and bg.get_device() == di):
        view, proj, campos, tanfov, bg = (t if t.device == dev else t.to(dev) for t in (view, proj, campos, tanfov, bg))
    if view.numel() == 16 and bg.numel() == 3:
        # one view (the per-camera loop of the reference): the kernels take pointers, so the camera blocks are handed on in
        # whatever shape they have -- four reshapes and a background view per call were ~8 us of host time
        V = 1
        view, proj, campos, tanfov, bg = _f32c(view), _f32c(proj), _f32c(campos), _f32c(tanfov), _f32c(bg)
        if proj.numel() != 16 or campos.numel() != 3 or tanfov.numel() != 2:
            raise _lib.GgsError("ggsplat: one view needs proj [16], campos [3], tanfov [2]")
    else:
        view = _f32c(view).reshape(-1, 16)
        V = view.shape[0]
        proj = _f32c(proj).reshape(V, 16)
        campos = _f32c(campos).reshape(V, 3)
        tanfov = _f32c(tanfov).reshape(V, 2)
        bg = _f32c(bg)
        if bg.numel() == 3:
            bg = bg.reshape(1, 3) if V == 1 else bg.reshape(1, 3).expand(V, 3).contiguous()
        else:
            bg = bg.reshape(V, 3).contiguous()
    K = shs.shape[1] if shs is not None else 0
    prm = _params(P, K, int(sh_degree), int(W), int(H), V, float(scale_modifier), int(bool(debug)))

    color = torch.empty(V, 3, H, W, device=dev, dtype=torch.float32)
    depth = torch.empty(V, H, W, device=dev, dtype=torch.float32)
    alpha = torch.empty(V, H, W, device=dev, dtype=torch.float32)
    radii = torch.empty(V, P, device=dev, dtype=torch.int32)
    return (dev, P, V, prm, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, view, proj, campos, tanfov,
            bg, color, depth, alpha, radii)


def forward_views(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, *, view, proj, campos,
                  tanfov, bg, W: int, H: int, sh_degree: int, scale_modifier: float = 1.0, debug: bool = False,
                  keep_state: bool = True, workspaces: Optional[StepWorkspaces] = None):
    """Rasterize V views.  view/proj [V,16], campos [V,3], tanfov [V,2], bg [V,3] or [3].
    Returns color [V,3,H,W], radii [V,P] int32, depth [V,H,W], alpha [V,H,W], state.
    workspaces: buffers from plan_step() for this shape (used as long as their capacity is the one the call runs with)."""
    L = lib()
    (dev, P, V, prm, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, view, proj, campos, tanfov, bg,
     color, depth, alpha, radii) = _prep_forward(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, view,
                                                 proj, campos, tanfov, bg, W, H, sh_degree, scale_modifier, debug)

    key = (dev.index, P, W, H, V)
    cap = _cap_hint.get(key, max(8 * P * V, 1 << 16))
    stream = _lib.stream_ptr(dev)
    geom = img = None
    ws = workspaces if (workspaces is not None and workspaces.key == key and workspaces.cap == cap) else None
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and key not in _cap_hint:
        raise _lib.GgsError("ggsplat: run this configuration eagerly once before capturing it into a graph "
                            "(the binning capacity is learnt from an eager call)")
    # the device pointers of the inputs, taken once: the backward of this forward is handed the same ones (ForwardState.bwd_args)
    p_in = (ptr(bg), ptr(means3D), ptr(shs), ptr(colors_precomp), ptr(opacities), ptr(scales), ptr(rotations),
            ptr(cov3D_precomp), ptr(view), ptr(proj), ptr(campos), ptr(tanfov))
    p_out = (ptr(color), ptr(depth), ptr(alpha), ptr(radii), stream)
    prm_ref = C.byref(prm)
    while True:
        gsz, isz, bsz = _workspace_sizes(L, prm, cap)
        if ws is not None and ws.cap == cap:          # (an overflow retry runs with a larger capacity and its own buffers)
            geom, img, binb = ws.geom, ws.img, ws.bin
        else:
            if geom is None:
                geom = torch.empty(gsz, device=dev, dtype=torch.uint8)
            if img is None or img.numel() < isz:
                img = torch.empty(isz, device=dev, dtype=torch.uint8)
            binb = torch.empty(bsz, device=dev, dtype=torch.uint8)
        p_ws = (geom.data_ptr(), binb.data_ptr(), cap, img.data_ptr())
        if capturing:       # one pass: static capacity, no host sync, overflow flag left on the device
            check(L.ggs_forward(prm_ref, *p_in, *p_ws, *p_out), "ggs_forward")
            n = -1
        else:
            # phase 1: preprocess + tile histogram + scan; its 16-byte header {num_rendered, overflow} is copied out behind it.
            # phase 2 (scatter + sort + composite) is queued right away, WITHOUT waiting for the header: every kernel of it is
            # guarded by the overflow word on the device, so with the capacity learnt from earlier calls it is simply the
            # render, and in the rare overflow case it composites nothing and the call is repeated with the exact size.  The
            # ONLY host sync of the call waits for the header copy (~10 us of GPU work in front of it -- the same point where
            # the upstream extension syncs to size its binning buffer) while the GPU is already compositing.
            host, ev = _pinned_header(dev), _header_event(dev)
            check(L.ggs_forward_spec(prm_ref, *p_in, *p_ws, *p_out, host.data_ptr(), ev.cuda_event), "ggs_forward_spec")
        # (what follows does not depend on the header: it runs while the GPU works through phase 1, the wait comes last)
        _pinned.last_bin = (binb, V, _lib.n_tiles(W, H), _tile_count_offset(L, prm, cap))
        st = None
        if keep_state:
            st = ForwardState()
            st.prm, st.bg, st.means3D, st.shs, st.colors, st.opac = prm, bg, means3D, shs, colors_precomp, opacities
            st.scales, st.rots, st.cov, st.view, st.proj, st.campos, st.tanfov = scales, rotations, cov3D_precomp, view, proj, campos, tanfov
            st.geom, st.bin, st.cap, st.img = geom, binb, cap, img
            # ggs_backward's leading arguments: (prm, bg, means3D, shs, colors, scales, rots, cov3D, view, proj, campos, tanfov,
            # geom, bin, capacity, img) -- the state holds the tensors, so the addresses stay valid
            st.bwd_args = (prm_ref,) + p_in[:4] + p_in[5:] + p_ws
        if capturing:
            _capture_headers.append(_header_of(binb))
            break
        ev.synchronize()
        n, overflow = host.tolist()
        if not overflow:
            _cap_hint[key] = max(cap if n * 2 <= cap else int(n * 2), 1 << 16)
            break
        cap = int(n * 1.25) + 1024             # n is exact: one retry is always enough
    if st is not None:
        st.num_rendered = n
    return color, radii, depth, alpha, st


class StagedForward:
    """One forward of V views whose three stages (ggs_forward_stages: count | bin | composite) the caller queues itself -- on
    one stream or on several it orders with events (ggsplat.batch.fwd_bwd_views(pipeline=2)).  No host sync: the call runs with
    the binning capacity learnt by earlier forward_views calls of the same shape (raises if there was none) and leaves its
    overflow word on the device (`header`); a forward that overflowed composites every tile as empty and every kernel
    behind it is guarded, so the caller checks `header[1]` once per step and falls back to forward_views."""

    def __init__(self, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, *, view, proj, campos, tanfov,
                 bg, W: int, H: int, sh_degree: int, scale_modifier: float = 1.0):
        L = lib()
        (dev, P, V, prm, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, view, proj, campos, tanfov, bg,
         color, depth, alpha, radii) = _prep_forward(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                                                     view, proj, campos, tanfov, bg, W, H, sh_degree, scale_modifier, False)
        key = (dev.index, P, W, H, V)
        if key not in _cap_hint:
            raise _lib.GgsError("ggsplat: run this configuration through forward_views once before staging it "
                                "(the binning capacity is learnt from an eager call)")
        cap = _cap_hint[key]
        gsz, isz, bsz = _workspace_sizes(L, prm, cap)
        new = (lambda n: torch.empty(n, device=dev, dtype=torch.uint8))
        geom, img, binb = new(gsz), new(isz), new(bsz)
        self.dev, self.outputs = dev, (color, radii, depth, alpha)
        self._args = (C.byref(prm), ptr(bg), ptr(means3D), ptr(shs), ptr(colors_precomp), ptr(opacities), ptr(scales),
                      ptr(rotations), ptr(cov3D_precomp), ptr(view), ptr(proj), ptr(campos), ptr(tanfov), ptr(geom),
                      ptr(binb), cap, ptr(img), ptr(color), ptr(depth), ptr(alpha), ptr(radii))
        self.header = _header_of(binb)
        self._last_bin = (binb, V, _lib.n_tiles(W, H), _tile_count_offset(L, prm, cap))
        st = self.state = ForwardState()
        st.prm, st.bg, st.means3D, st.shs, st.colors, st.opac = prm, bg, means3D, shs, colors_precomp, opacities
        st.scales, st.rots, st.cov, st.view, st.proj, st.campos, st.tanfov = scales, rotations, cov3D_precomp, view, proj, campos, tanfov
        st.geom, st.bin, st.cap, st.img, st.num_rendered = geom, binb, cap, img, -1
        st.bwd_args = self._args[:5] + self._args[6:17]
        if torch.cuda.is_current_stream_capturing():
            _capture_headers.append(self.header)

    COUNT, BIN, COMPOSITE = 1, 2, 4

    @property
    def tile_count(self) -> Optional[torch.Tensor]:
        """List lengths of the 16x16 tiles of THIS forward, int32 [V, T] (what last_tile_count() returns for a forward_views call)."""
        if _lib.tile_size() != (16, 16):
            return None
        binb, V, T, off = self._last_bin
        return binb[off:off + V * T * 4].view(torch.int32).reshape(V, T)

    def make_current(self) -> None:
        """Make this forward the calling thread's "most recent forward": last_header() / last_tile_count() then describe IT.
        With stages of several forwards in flight the most recently ISSUED one is not the one a loss callback is about to
        consume (ggsplat.batch sets it right before dL_dcolor_fn(i))."""
        _pinned.last_bin = self._last_bin

    def run(self, stages: int) -> None:
        """Queue the given stages on PyTorch's current stream."""
        check(lib().ggs_forward_stages(int(stages), *self._args, _lib.stream_ptr(self.dev)), "ggs_forward_stages")


def new_grads(P: int, K: int, has_shs: bool, has_cov: bool, dev) -> Dict[str, torch.Tensor]:
    """The (uninitialised) gradient tensors ggs_backward writes for P Gaussians in the given input mode."""
    new = (lambda *s: torch.empty(*s, device=dev, dtype=torch.float32))
    g = {"means3D": new(P, 3), "opacities": new(P, 1)}
    if has_shs:
        g["shs"] = new(P, K, 3)
    else:
        g["colors_precomp"] = new(P, 3)
    if has_cov:
        g["cov3D_precomp"] = new(P, 6)
    else:
        g["scales"], g["rotations"] = new(P, 3), new(P, 4)
    return g


def backward_views(st: ForwardState, dL_dcolor, dL_ddepth=None, dL_dalpha=None, want_means2D: bool = True,
                   out: Optional[Dict[str, torch.Tensor]] = None, accumulate: bool = False,
                   scratch: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Gradients summed over the V views of `st` (dL_dmeans2D stays per view).  scratch: plan_step().scratch or None."""
    L = lib()
    prm = st.prm
    dev = st.means3D.device
    P, K, V = prm.P, prm.K, prm.n_views
    dL_dcolor = _f32c(dL_dcolor)
    dL_ddepth, dL_dalpha = _f32c(dL_ddepth), _f32c(dL_dalpha)
    g = out if out is not None else {}
    if "means3D" not in g:
        g.update(new_grads(P, K, st.shs is not None, st.cov is not None, dev))
        accumulate = False
    if want_means2D and "means2D" not in g:
        g["means2D"] = torch.empty(V, P, 3, device=dev, dtype=torch.float32)
    k = (P, K, V)
    nbytes = _bwd_scratch.get(k)
    if nbytes is None:
        if len(_bwd_scratch) > 256:
            _bwd_scratch.clear()
        nbytes = _bwd_scratch[k] = int(L.ggs_backward_scratch_bytes(C.byref(prm)))
    if scratch is None or scratch.numel() < nbytes:
        scratch = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    gg = g.get
    # (prm, inputs, workspaces): the addresses the forward of this state was called with
    check(L.ggs_backward(*st.bwd_args, ptr(dL_dcolor), ptr(dL_ddepth),
                         ptr(dL_dalpha), scratch.data_ptr(), ptr(gg("means2D") if want_means2D else None),
                         ptr(g["means3D"]), ptr(g["opacities"]), ptr(gg("shs")), ptr(gg("colors_precomp")),
                         ptr(gg("scales")), ptr(gg("rotations")), ptr(gg("cov3D_precomp")),
                         int(bool(accumulate)), _stream_ptr(dev)), "ggs_backward")
    return g


def _alias(t: torch.Tensor, shape) -> torch.Tensor:
    """A tensor over the same storage as contiguous `t` with a new shape that autograd does not see as a view."""
    return t.new_empty(0).set_(t.untyped_storage(), t.storage_offset(), tuple(shape))


def bin_sections(st: ForwardState) -> Dict[str, torch.Tensor]:
    """Typed views into the binning buffer of a forward (tests / debugging)."""
    off = (C.c_size_t * 8)()
    check(lib().ggs_bin_layout(C.byref(st.prm), st.cap, off), "ggs_bin_layout")
    V = st.prm.n_views
    T = _lib.n_tiles(st.prm.W, st.prm.H)
    b = st.bin

    def sec(i, nbytes, dt):
        return b[off[i]:off[i] + nbytes].view(dt)

    return dict(header=sec(0, 16, torch.int64), tile_count=sec(1, V * T * 4, torch.int32).reshape(V, T),
                tile_offset=sec(3, V * T * 4, torch.int32).reshape(V, T), view_base=sec(4, V * 8, torch.int64),
                keys=sec(5, st.cap * 8, torch.int64), ids=sec(6, st.cap * 4, torch.int32),
                # work items in launch order: behind view_base (256-byte aligned sections, csrc/ggs_common.h ggs_bin_layout)
                order=b[off[4] + ((V * 8 + 255) & ~255):off[4] + ((V * 8 + 255) & ~255) + V * T * 4].view(torch.int32))


def img_sections(st: ForwardState) -> Dict[str, torch.Tensor]:
    """The image workspace of a forward (tests / debugging): final_T [V,H,W] f32 and n_contrib [V,H,W] i32 (1-based list
    position of the last contributor).  The kernels leave the workspace of EMPTY tiles undefined -- nothing on the device reads
    it -- so the copies returned here carry the values an empty tile has by definition there (final_T = 1, n_contrib = 0)."""
    V, H, W = st.prm.n_views, st.prm.H, st.prm.W
    n = V * H * W * 4
    off = (n + 255) & ~255
    final_T = st.img[:n].view(torch.float32).reshape(V, H, W).clone()
    n_contrib = st.img[off:off + n].view(torch.int32).reshape(V, H, W).clone()
    tw, th = _lib.tile_size()
    gx, gy = (W + tw - 1) // tw, (H + th - 1) // th
    empty = (bin_sections(st)["tile_count"].reshape(V, gy, gx) == 0)
    empty = empty.repeat_interleave(th, 1).repeat_interleave(tw, 2)[:, :H, :W]
    final_T[empty] = 1.0
    n_contrib[empty] = 0
    return dict(final_T=final_T, n_contrib=n_contrib)


class _RasterizeGaussians(torch.autograd.Function):
    """Single-view autograd op with the reference extension's argument order."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        dev = means3D.device
        tanfov = getattr(settings, "tanfov", None)      # device [1,2] of a static (graph-captured) camera
        if tanfov is None:
            tanfov = tanfov_tensor(settings.tanfovx, settings.tanfovy, dev)
        color, radii, depth, alpha, st = forward_views(
            means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp,
            view=settings.viewmatrix, proj=settings.projmatrix, campos=settings.campos, tanfov=tanfov,
            bg=settings.bg, W=settings.image_width, H=settings.image_height, sh_degree=settings.sh_degree,
            scale_modifier=settings.scale_modifier, debug=settings.debug, keep_state=True)
        ctx.set_materialize_grads(False)      # unused depth / alpha outputs -> None, not zeros
        ctx.st = st
        # The backward recomputes cov3D / SH colours from the INPUTS (they are not copied), so an in-place edit between
        # forward and backward would silently change the gradients.  save_for_backward would also hold the outputs'
        # graph alive for callers that mutate them; recording the version counters gives the same protection: the
        # upstream extension raises autograd's "modified by an inplace operation" error in that situation, so do we.
        # (Only the tensors the native backward re-reads; `opacities` is not among them -- the backward takes the opacity
        # from the forward's records -- and upstream does not save it either.)
        ctx.in_versions = [(n, t, t._version) for n, t in (("means3D", means3D), ("sh", sh), ("colors_precomp", colors_precomp),
                                                            ("scales", scales), ("rotations", rotations),
                                                            ("cov3Ds_precomp", cov3Ds_precomp)) if t is not None]
        ctx.m2d_shape = means2D.shape
        # Outputs are never saved: callers mutate them in place (ssim does `img1 *= mask`,
        # utils/loss_utils.py:44-46).  For the same reason they must not be VIEWS created inside this
        # Function (autograd refuses in-place ops on those): alias the storage with fresh tensors.
        color1 = _alias(color, (3, settings.image_height, settings.image_width))
        radii1 = _alias(radii, (radii.shape[1],))
        ctx.mark_non_differentiable(radii1)
        return color1, radii1, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        st = ctx.st
        for name, t, ver in ctx.in_versions:
            if t._version != ver:
                raise RuntimeError(f"one of the variables needed for gradient computation has been modified by an inplace "
                                   f"operation: rasterizer input `{name}` is at version {t._version}; expected version {ver}")
        H, W = st.prm.H, st.prm.W
        if g_color is None:
            g_color = torch.zeros(3, H, W, device=st.means3D.device)
        g = backward_views(st, g_color.reshape(1, 3, H, W), None if g_depth is None else g_depth.reshape(1, H, W),
                           None if g_alpha is None else g_alpha.reshape(1, H, W), want_means2D=True)
        return (g["means3D"], g["means2D"][0].reshape(ctx.m2d_shape), g.get("shs"), g.get("colors_precomp"),
                g["opacities"].reshape(st.opac.shape), g.get("scales"), g.get("rotations"), g.get("cov3D_precomp"), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, settings)
