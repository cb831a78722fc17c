"""Multi-view batched forward+backward -- the throughput mode of the registration /
appearance inner step (s2_registration.py:213-334, s3_appearance.py:107-149).

The reference renders ONE camera per optimisation step.  For fixed parameters the views are
independent, so this module renders a shard of views per launch set (grid dimension = view),
accumulates the parameter gradients over all of them on the device, and (multi-GPU) sums the flat
gradient bucket with a single RCCL all-reduce (ggsplat.dist).  Semantics: one optimiser step per
batch of views instead of per view -- see DESIGN.md "multi-GPU".
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import rasterizer as R


_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    """The second stream of the pipelined form (one per device, created once: a stream maps to a hardware queue)."""
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _side_streams.get(i)
    if s is None:
        s = _side_streams[i] = torch.cuda.Stream(device=dev)
    return s


def fwd_bwd_views(inputs: Dict[str, torch.Tensor], cams: Dict[str, torch.Tensor], *, bg: torch.Tensor, W: int, H: int,
                  sh_degree: int, dL_dcolor_fn, chunk: int = 32, keep_images: bool = False,
                  want_means2D: bool = False, pipeline: int = 0) -> Dict[str, torch.Tensor]:
    """inputs: activated rasterizer inputs {means3D, opacities, shs|colors_precomp, scales+rotations|cov3D_precomp}.
    cams: stacked {view, proj, campos, tanfov} for this rank's views.
    dL_dcolor_fn(v0, v1, color[v0:v1]) -> dL/dcolor [v1-v0,3,H,W] (the loss backward of those views).
    Returns gradients w.r.t. the activated inputs, summed over all views (+ stats).

    pipeline (two or more launch sets; 0 = serial, the default): launch sets software-pipelined over a SECOND stream -- for
    fixed parameters the sets only meet in the gradient accumulators, which the backward chain owns.  1: the whole backward of
    set i beside the whole forward of set i + 1; 2: count + bin of set i + 1 in front of the backward of set i, only its
    compositing beside it (_fwd_bwd_views_staged).  Both work eagerly and inside a stream capture (the second stream joins the
    capture through the events and is joined back before returning: the captured graph has parallel branches).  Same kernels,
    same order of the backward launches; the results equal the serial form up to the order of the float atomics inside the
    render backward (run-to-run noise of the serial form itself, ~1e-7).  MEASURED on MI355X (profiles/r05_pipeline_overlap.md,
    r05_pipeline_sweep.txt): with 1 the two render kernels never meet (the forward's 256-thread kernels are starved beside the
    backward's flood of one-wave workgroups) but launch gaps and small kernels hide under the other chain: +0.6 ... +1.7 %, bench.py's
    default on one GPU; with 2 the render kernels interleave and take each other's issue slots: -1 %."""
    V = cams["view"].shape[0]
    dev = inputs["means3D"].device
    if int(pipeline) == 2 and V > chunk and dev.type == "cuda" and not keep_images:
        r = _fwd_bwd_views_staged(inputs, cams, bg=bg, W=W, H=H, sh_degree=sh_degree, dL_dcolor_fn=dL_dcolor_fn, chunk=chunk,
                                  want_means2D=want_means2D)
        if r is not None:
            return r
        pipeline = 0                        # no learnt capacity yet (first call of this shape): the serial form learns it
    pipeline = bool(pipeline) and V > chunk and dev.type == "cuda"
    return _fwd_bwd_views_serial(inputs, cams, bg=bg, W=W, H=H, sh_degree=sh_degree, dL_dcolor_fn=dL_dcolor_fn, chunk=chunk,
                                 keep_images=keep_images, want_means2D=want_means2D, pipeline=pipeline)


def _fwd_bwd_views_serial(inputs, cams, *, bg, W, H, sh_degree, dL_dcolor_fn, chunk, keep_images, want_means2D, pipeline):
    """pipeline False: launch sets one after the other; True: the backward of set i on the second stream (fwd_bwd_views)."""
    V = cams["view"].shape[0]
    dev = inputs["means3D"].device
    grads: Optional[Dict[str, torch.Tensor]] = None
    n_total = 0
    images: List[torch.Tensor] = []
    # dL/dmeans2D is per view: one [V,P,3] tensor, every launch set writes its own rows (no concatenation afterwards)
    m2d = torch.empty(V, inputs["means3D"].shape[0], 3, device=dev, dtype=torch.float32) if want_means2D else None
    main = side = None
    keep: List = []                        # pipelined: everything the second stream still reads stays referenced until the join
    if pipeline:
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        P = inputs["means3D"].shape[0]
        shs = inputs.get("shs")
        # the accumulators are allocated on the caller's stream (they outlive the second stream's work)
        grads = R.new_grads(P, shs.shape[1] if shs is not None else 0, shs is not None, inputs.get("cov3D_precomp") is not None, dev)
        side.wait_stream(main)
    first = True
    try:
        for v0 in range(0, V, chunk):
            v1 = min(V, v0 + chunk)
            color, radii, depth, alpha, st = R.forward_views(
                inputs["means3D"], inputs["opacities"], inputs.get("shs"), inputs.get("colors_precomp"),
                inputs.get("scales"), inputs.get("rotations"), inputs.get("cov3D_precomp"),
                view=cams["view"][v0:v1], proj=cams["proj"][v0:v1], campos=cams["campos"][v0:v1],
                tanfov=cams["tanfov"][v0:v1], bg=bg, W=W, H=H, sh_degree=sh_degree)
            n_total += st.num_rendered
            dL = dL_dcolor_fn(v0, v1, color)
            if want_means2D and grads is not None:
                grads["means2D"] = m2d[v0:v1]
            if pipeline:
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    R.backward_views(st, dL, want_means2D=want_means2D, out=grads, accumulate=not first)
                keep.append((st, color, radii, depth, alpha, dL))
            elif grads is None:
                grads = R.backward_views(st, dL, want_means2D=want_means2D, out={"means2D": m2d[v0:v1]} if want_means2D else None)
            else:
                R.backward_views(st, dL, want_means2D=want_means2D, out=grads, accumulate=True)
            first = False
            if keep_images:
                images.append(color)
            del st
    finally:
        if pipeline:                        # also when the callback or a launch raised: the second stream is joined BEFORE the
            main.wait_stream(side)          # tensors it reads are released
            keep.clear()
    grads = grads or {}
    grads["num_rendered"] = n_total
    if want_means2D:
        grads["means2D"] = m2d                 # [V,P,3], every view of every launch set
    if keep_images:
        grads["images"] = torch.cat(images)
    return grads


def _fwd_bwd_views_staged(inputs, cams, *, bg, W, H, sh_degree, dL_dcolor_fn, chunk, want_means2D):
    """pipeline=2: the launch sets software-pipelined at STAGE granularity (rasterizer.StagedForward).  Whole-forward pipelining
    (pipeline=1) does not overlap anything on this part: a kernel of 256-thread workgroups (preprocess, scatter, sort) queued
    beside a flood of one-wave workgroups (the render backward) is starved until the flood drains (profiles/r05_pipeline_
    overlap.md).  Here everything with wide workgroups of set i + 1 -- count and bin -- is queued IN FRONT of the backward of set
    i on the caller's stream, and only the compositing of set i + 1 (one-wave workgroups, like the backward's) runs beside it
    on the second stream.  Needs the binning capacity of this shape (learnt by an earlier serial call): returns None if there
    is none.  No host sync inside; num_rendered is read from the device headers at the end.  If a set overflowed its capacity
    (seen only then) the call returns None and fwd_bwd_views repeats the step serially: dL_dcolor_fn is then invoked a SECOND
    time for every set -- a callback that accumulates (loss sums, logs) must tolerate that or use pipeline <= 1."""
    V = cams["view"].shape[0]
    dev = inputs["means3D"].device
    bounds = [(v0, min(V, v0 + chunk)) for v0 in range(0, V, chunk)]
    P = inputs["means3D"].shape[0]
    shs = inputs.get("shs")

    def staged(v0, v1):
        return R.StagedForward(inputs["means3D"], inputs["opacities"], shs, inputs.get("colors_precomp"), inputs.get("scales"),
                               inputs.get("rotations"), inputs.get("cov3D_precomp"), view=cams["view"][v0:v1],
                               proj=cams["proj"][v0:v1], campos=cams["campos"][v0:v1], tanfov=cams["tanfov"][v0:v1], bg=bg,
                               W=W, H=H, sh_degree=sh_degree)
    try:
        cur = staged(*bounds[0])            # (at most two sets are alive at a time: the one being differentiated and the next)
    except R._lib.GgsError:
        return None
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    grads = R.new_grads(P, shs.shape[1] if shs is not None else 0, shs is not None, inputs.get("cov3D_precomp") is not None, dev)
    m2d = torch.empty(V, P, 3, device=dev, dtype=torch.float32) if want_means2D else None
    hdrs = torch.empty(len(bounds), 2, dtype=torch.int64, device=dev)      # {num_rendered, overflow} of every set
    SF = R.StagedForward
    side.wait_stream(main)
    try:
        cur.run(SF.COUNT | SF.BIN | SF.COMPOSITE)
        composited = None                   # event: compositing of the set whose backward comes next (None: it ran on `main`)
        for i, (v0, v1) in enumerate(bounds):
            nxt = staged(*bounds[i + 1]) if i + 1 < len(bounds) else None
            ev_next = None
            if nxt is not None:
                nxt.run(SF.COUNT | SF.BIN)  # wide-workgroup kernels of set i + 1: in front of the backward of set i
                front = torch.cuda.Event()
                front.record(main)
                with torch.cuda.stream(side):   # its compositing: beside that backward
                    side.wait_event(front)
                    nxt.run(SF.COMPOSITE)
                    ev_next = torch.cuda.Event()
                    ev_next.record(side)
            if composited is not None:
                main.wait_event(composited)
            color = cur.outputs[0]
            cur.make_current()              # rasterizer.last_header() / last_tile_count() inside the callback: THIS set's tables
            dL = dL_dcolor_fn(v0, v1, color)
            if want_means2D:
                grads["means2D"] = m2d[v0:v1]
            R.backward_views(cur.state, dL, want_means2D=want_means2D, out=grads, accumulate=i > 0)
            hdrs[i].copy_(cur.header)
            # set i is done on `main`, which also waited for its compositing on `side`: its buffers go back to the allocator
            # (blocks of the caller's stream, reused by work queued behind this point)
            cur, composited = nxt, ev_next
    finally:
        main.wait_stream(side)              # also when the callback or a launch raised: nothing of this call is left on `side`
    grads.pop("means2D", None)
    if want_means2D:
        grads["means2D"] = m2d
    if torch.cuda.is_current_stream_capturing():
        grads["num_rendered"] = -1
    else:
        hdr = hdrs.cpu()                    # the one host sync of the step
        if bool((hdr[:, 1] != 0).any()):
            R.grow_capacity(2.0)
            return None                     # a set overflowed its binning capacity: the serial form re-sizes per call
        grads["num_rendered"] = int(hdr[:, 0].sum())
    return grads


def model_fwd_bwd_views(model, cams: Dict[str, torch.Tensor], *, bg: torch.Tensor, W: int, H: int, dL_dcolor_fn,
                        chunk: int = 32, pipeline: int = 0, want_means2D: bool = False) -> Dict[str, torch.Tensor]:
    """fwd_bwd_views for a MeshGaussianModel, gradients w.r.t. its PARAMETERS: mesh binding -> render forward + backward of
    every view -> mesh-binding backward, sigmoid backward of the opacities, assembled from the C entry points without the
    autograd graph (same kernels, ~20 small PyTorch launches fewer per step -- what a rank of an 8-way sharded step, 20 views,
    notices).  Returns {"flat": the gradient bucket [mesh.v | _xyz | f_dc | f_rest | opacity | scaling | rotation] (the
    order of model.parameters(), what ggsplat.dist all-reduces), "num_rendered"}."""
    import ctypes as C

    from ._lib import check, lib, ptr
    g = model
    L = lib()
    dev = g._xyz.device
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P, Fn = g._xyz.shape[0], g.mesh.f.shape[0]
    with torch.no_grad():
        verts, faces, binding, bary = g.mesh.v, g.mesh.f, g.binding, getattr(g, "gs_bc", None)
        xyz, scaling, rot = torch.empty_like(g._xyz), torch.empty_like(g._scaling), torch.empty_like(g._rotation)
        check(L.ggs_mesh_bind_forward(P, Fn, ptr(verts), ptr(faces), ptr(binding), ptr(g._xyz), ptr(g._scaling),
                                      ptr(g._rotation), ptr(bary), ptr(xyz), ptr(scaling), ptr(rot), stream),
              "ggs_mesh_bind_forward")
        opacity = torch.sigmoid(g._opacity)
        K = 1 + g._features_rest.shape[1]
        shs = g._features_dc if K == 1 else torch.cat((g._features_dc, g._features_rest), dim=1)
        gr = fwd_bwd_views(dict(means3D=xyz, scales=scaling, rotations=rot, opacities=opacity, shs=shs), cams, bg=bg, W=W,
                           H=H, sh_degree=g.active_sh_degree, chunk=chunk, dL_dcolor_fn=dL_dcolor_fn, pipeline=pipeline,
                           want_means2D=want_means2D)
        d_verts = torch.zeros_like(verts)
        d_xyz, d_ls, d_rr = torch.empty_like(g._xyz), torch.empty_like(g._scaling), torch.empty_like(g._rotation)
        check(L.ggs_mesh_bind_backward(P, Fn, ptr(verts), ptr(faces), ptr(binding), ptr(g._xyz), ptr(g._scaling),
                                       ptr(g._rotation), ptr(bary), ptr(gr["means3D"]), ptr(gr["scales"]),
                                       ptr(gr["rotations"]), ptr(d_verts), ptr(d_xyz), ptr(d_ls), ptr(d_rr), stream),
              "ggs_mesh_bind_backward")
        d_op = torch.ops.aten.sigmoid_backward(gr["opacities"].reshape(opacity.shape), opacity)
        gs = gr["shs"]
        parts = [d_verts, d_xyz, gs if K == 1 else gs[:, :1], gs[:, 1:], d_op, d_ls, d_rr]
        flat = torch.cat([t.reshape(-1) for t in parts])
    out = {"flat": flat, "num_rendered": gr["num_rendered"]}
    if want_means2D:
        out["means2D"] = gr["means2D"]          # [V,P,3] screen-space gradients of every view (densification statistics)
    return out
