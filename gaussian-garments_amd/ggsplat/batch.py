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


def fwd_bwd_views(inputs: Dict[str, torch.Tensor], cams: Dict[str, torch.Tensor], *, bg: torch.Tensor, W: int, H: int,
                  sh_degree: int, dL_dcolor_fn, chunk: int = 32, keep_images: bool = False,
                  want_means2D: bool = False) -> Dict[str, torch.Tensor]:
    """inputs: activated rasterizer inputs {means3D, opacities, shs|colors_precomp, scales+rotations|cov3D_precomp}.
    cams: stacked {view, proj, campos, tanfov} for this rank's views.
    dL_dcolor_fn(v0, v1, color[v0:v1]) -> dL/dcolor [v1-v0,3,H,W] (the loss backward of those views).
    Returns gradients w.r.t. the activated inputs, summed over all views (+ stats)."""
    V = cams["view"].shape[0]
    grads: Optional[Dict[str, torch.Tensor]] = None
    n_total = 0
    images: List[torch.Tensor] = []
    m2d: List[torch.Tensor] = []          # dL/dmeans2D is per view: one [V_chunk,P,3] block per launch set
    for v0 in range(0, V, chunk):
        v1 = min(V, v0 + chunk)
        color, radii, depth, alpha, st = R.forward_views(
            inputs["means3D"], inputs["opacities"], inputs.get("shs"), inputs.get("colors_precomp"),
            inputs.get("scales"), inputs.get("rotations"), inputs.get("cov3D_precomp"),
            view=cams["view"][v0:v1], proj=cams["proj"][v0:v1], campos=cams["campos"][v0:v1],
            tanfov=cams["tanfov"][v0:v1], bg=bg, W=W, H=H, sh_degree=sh_degree)
        n_total += st.num_rendered
        dL = dL_dcolor_fn(v0, v1, color)
        if grads is None:
            grads = R.backward_views(st, dL, want_means2D=want_means2D)
        else:
            grads.pop("means2D", None)
            R.backward_views(st, dL, want_means2D=want_means2D, out=grads, accumulate=True)
        if want_means2D:
            m2d.append(grads["means2D"])
        if keep_images:
            images.append(color)
        del st
    grads = grads or {}
    grads["num_rendered"] = n_total
    if want_means2D and m2d:
        grads["means2D"] = m2d[0] if len(m2d) == 1 else torch.cat(m2d)       # [V,P,3], every view of every chunk
    if keep_images:
        grads["images"] = torch.cat(images)
    return grads


def model_fwd_bwd_views(model, cams: Dict[str, torch.Tensor], *, bg: torch.Tensor, W: int, H: int, dL_dcolor_fn,
                        chunk: int = 32) -> Dict[str, torch.Tensor]:
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
                           H=H, sh_degree=g.active_sh_degree, chunk=chunk, dL_dcolor_fn=dL_dcolor_fn)
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
    return {"flat": flat, "num_rendered": gr["num_rendered"]}
