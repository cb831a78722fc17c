"""Inner optimisation steps of stage 2 (registration) and stage 3 (appearance) -- host-side counterparts
of s2_registration.py:238-327 and s3_appearance.py:115-147, on the HIP rasterizer and the fused mesh
binding.  Same order of operations and the same loss composition; what is NOT here (out of scope,
SURVEY section 2): cloth energies, the StyleUNet (a caller-supplied `net` stands in for it), data loading, logging.
Density control (s2_registration.py:310-322) is ggsplat.densify; a captured step re-captures itself when it changed P."""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Callable, Dict, Optional

import torch
import torch.nn.functional as F

from . import rasterizer as R
from .adam import GraphAdam
from .loss import fused_photometric_loss, l1_loss, ssim
from ._lib import stream_ptr
from .render import render

DEFAULT_OPT = SimpleNamespace(                      # arguments/__init__.py:74-116 (the values the steps read)
    position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30_000,
    feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
    percent_dense=0.01, lambda_dssim=0.2, lambda_xyz=1e-2, threshold_xyz=1.0, lambda_scale=1.0,
    threshold_scale=0.6, threshold_opacity=0.75, lambda_opacity=0.01, only_foreground_loss=True)
DEFAULT_PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)


def _photometric(image, gt_image, m, lam, fused: bool, tile_count=None):
    """The two image terms of both loops: L1 * (1 - lambda) and 1 - SSIM * lambda.  fused=True: one pair of HIP
    kernels (ggs_photometric_*) in their region-of-interest form -- the image is the output of the render() call just
    made on this thread, and its gradient is only needed on the pixels of tiles that have a splat list; fused=False: the
    reference's PyTorch composition (loss.py), which also masks `image` and `gt_image` in place like the reference does."""
    if fused:
        # tile_count: the list lengths of the forward that PRODUCED `image` (render()'s pkg["tile_count"]); never the
        # thread's "most recent forward" looked up here -- another render of the same size in between (an evaluation view)
        # would hand over lists of the wrong image and silently zero the gradient on tiles that do have splats
        tc = tile_count
        if tc is not None and tc.shape != (1, ((image.shape[-2] + 15) // 16) * ((image.shape[-1] + 15) // 16)):
            tc = None
        return fused_photometric_loss(image, gt_image, m, lam, tile_count=tc)
    return l1_loss(image, gt_image, m) * (1.0 - lam), 1.0 - ssim(image, gt_image, m) * lam


class _HingeRegularisers(torch.autograd.Function):
    """The two hinge terms of the first-frame template (s2_registration.py:262-265) as ONE autograd node on the kernels of
    csrc/ggs_regaux.hip: loss_xyz = lambda_xyz mean_vis relu(|_xyz| - thr), loss_scale = lambda_scale mean_vis |relu(exp(_scaling)
    - thr)|_2, means over the Gaussians with radii > 0.  The kernels produce values AND gradients in two launches; the PyTorch
    composition is ~12 forward and ~15 backward nodes, most of the eager step's autograd time.  Used by registration_step when
    fused_loss=True; tested against the composition (tests/test_gpu_inner_step.py)."""

    @staticmethod
    def forward(ctx, xyz, log_scaling, radii, thr_xyz, lam_xyz, thr_scale, lam_scale):
        import ctypes as C
        from ._lib import check, lib, ptr
        dev = xyz.device
        P = xyz.shape[0]
        x, ls = xyz.detach(), log_scaling.detach()
        x = x if x.is_contiguous() else x.contiguous()
        ls = ls if ls.is_contiguous() else ls.contiguous()
        d_xyz, d_ls = torch.zeros_like(x), torch.zeros_like(ls)
        out = torch.empty(3, device=dev, dtype=torch.float32)
        scratch = torch.empty(4, device=dev, dtype=torch.float32)
        r = radii if (radii.dtype is torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
        check(lib().ggs_registration_aux(P, ptr(x), ptr(ls), ptr(r), None, None, None, None, float(thr_xyz), float(lam_xyz),
                                         float(thr_scale), float(lam_scale), ptr(d_xyz), ptr(d_ls), None, None, None, ptr(out),
                                         ptr(scratch), None, stream_ptr(dev)),
              "ggs_registration_aux")
        ctx.saved = (d_xyz, d_ls)
        ctx.mark_non_differentiable(radii)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_xyz, g_scale):
        d_xyz, d_ls = ctx.saved
        return (None if g_xyz is None else d_xyz * g_xyz, None if g_scale is None else d_ls * g_scale, None, None, None, None, None)


def _densification_stats_fused(gaussians, vsp_grad, radii, hdr):
    """max_radii2D / xyz_gradient_accum / denom of the visible Gaussians in one launch (ggs_registration_aux, statistics only);
    hdr: the forward's {num_rendered, overflow} words when a guarded (replayable) step must leave them alone on overflow."""
    import ctypes as C
    from ._lib import check, lib, ptr
    g = gaussians
    dev = g._xyz.device
    r = radii if (radii.dtype is torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
    gr = vsp_grad if vsp_grad.is_contiguous() else vsp_grad.contiguous()
    check(lib().ggs_registration_aux(g._xyz.shape[0], None, None, ptr(r), ptr(gr), None, None, None, 0.0, 0.0, 0.0, 0.0, None, None,
                                     ptr(g.max_radii2D), ptr(g.xyz_gradient_accum), ptr(g.denom), None, None,
                                     None if hdr is None else ptr(hdr[1:2]), stream_ptr(dev)),
          "ggs_registration_aux")


def registration_step(gaussians, viewpoint_cam, gt_image, mask, bg, opt=DEFAULT_OPT, pipe=DEFAULT_PIPE,
                      first_frame_template: bool = True, track_densification: bool = True,
                      optimizer_step: bool = True, fused_loss: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of the s2 loop: update_face_coor -> render -> L1 (1 - lambda) + (1 - ssim lambda)
    [+ xyz / scale hinges on the first template frame] -> backward -> densification stats -> Adam step."""
    gaussians.update_face_coor()
    pkg = render(viewpoint_cam, gaussians, pipe, bg)
    image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
    m = mask if opt.only_foreground_loss else None
    l_img, l_ssim = _photometric(image, gt_image, m, opt.lambda_dssim, fused_loss, pkg.get("tile_count"))
    loss_dict = {"img": l_img, "ssim": l_ssim}
    if first_frame_template and fused_loss and gaussians._xyz.is_cuda:
        loss_dict["xyz"], loss_dict["scale"] = _HingeRegularisers.apply(
            gaussians._xyz, gaussians._scaling, radii, opt.threshold_xyz, opt.lambda_xyz, opt.threshold_scale, opt.lambda_scale)
    elif first_frame_template:
        # means over the visible Gaussians (s2_registration.py:262-265 index with [visibility_filter]); written with
        # masks so that nothing depends on a host-side count -- same values, and the step stays graph-capturable
        visf = vis.to(image.dtype)
        n_vis = visf.sum()
        loss_dict["xyz"] = (F.relu(gaussians._xyz.norm(dim=1) - opt.threshold_xyz) * visf).sum() / n_vis * opt.lambda_xyz
        loss_dict["scale"] = (F.relu(gaussians.scaling_activation(gaussians._scaling) - opt.threshold_scale
                                     ).norm(dim=1) * visf).sum() / n_vis * opt.lambda_scale
    loss = sum(loss_dict.values())
    loss.backward()
    with torch.no_grad():
        graph_opt = isinstance(gaussians.optimizer, GraphAdam)
        hdr = R.last_header() if graph_opt else None            # {num_rendered, overflow} of this step's forward
        fused_stats = (fused_loss and gaussians.max_radii2D.dtype is torch.float32 and gaussians.max_radii2D.is_contiguous()
                       and vsp.grad is not None and vsp.grad.shape[0] == gaussians.max_radii2D.shape[0])
        if first_frame_template and track_densification and fused_stats:
            _densification_stats_fused(gaussians, vsp.grad, radii, hdr)
        elif first_frame_template and track_densification:
            mr = gaussians.max_radii2D
            new_mr = torch.where(vis, torch.max(mr, radii.to(mr.dtype)), mr)
            ok = None
            if hdr is not None:                                  # a replayed step that overflowed changes nothing
                ok = (hdr[1] == 0).to(gaussians.denom.dtype)
                new_mr = torch.where(hdr[1] == 0, new_mr, mr)
            mr.copy_(new_mr)
            gaussians.add_densification_stats(vsp, vis, ok=ok)
        if optimizer_step and gaussians.optimizer is not None:
            if graph_opt:
                gaussians.optimizer.step(guard=hdr[1:2])
            else:
                gaussians.optimizer.step()
            gaussians.optimizer.zero_grad()
    loss_dict["loss"] = loss.detach()
    loss_dict["render_pkg"] = pkg
    return loss_dict


def appearance_step(gaussians, net: Callable, viewpoint_cam, gt_image, mask, bg, optimizer=None,
                    opt=DEFAULT_OPT, pipe=DEFAULT_PIPE, fused_loss: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of the s3 loop.  `net(gaussians, cam) -> (xyz_offset [P,3], sh_offset [P,K,3], vis_mask [P])`
    stands in for AvatarNet.forward (scene/avatar_net.py:58-87): it sets local_xyz = _xyz + offset and
    shs = get_features + offset, then render(..., vis_mask=vis_mask) and the five-term loss."""
    gaussians.update_face_coor()
    xyz_off, sh_off, vis_mask = net(gaussians, viewpoint_cam)
    gaussians.local_xyz = gaussians._xyz + xyz_off
    gaussians.shs = gaussians.get_features + sh_off
    pkg = render(viewpoint_cam, gaussians, pipe, bg, vis_mask=vis_mask)
    image = pkg["render"]
    m = mask if opt.only_foreground_loss else None
    l_img, l_ssim = _photometric(image, gt_image, m, opt.lambda_dssim, fused_loss, pkg.get("tile_count"))
    loss_dict = {"img": l_img, "ssim": l_ssim,
                 "xyz": F.relu(gaussians.local_xyz.norm(dim=1) - opt.threshold_xyz).mean() * opt.lambda_xyz,
                 "scale": F.relu(gaussians.scaling_activation(gaussians._scaling) - opt.threshold_scale
                                 ).norm(dim=1).mean() * opt.lambda_scale,
                 "opacity": F.relu(opt.threshold_opacity - gaussians.get_opacity).mean() * opt.lambda_opacity}
    loss = sum(loss_dict.values())
    loss.backward()
    if optimizer is not None:
        with torch.no_grad():
            if isinstance(optimizer, GraphAdam):
                optimizer.step(guard=R.last_header()[1:2])      # void when this step's forward overflowed (replay)
            else:
                optimizer.step()
            optimizer.zero_grad()
    loss_dict["loss"] = loss.detach()
    loss_dict["render_pkg"] = pkg
    return loss_dict


_BLK_BYTES = 184            # the per-iteration parameter block of the captured steps (see GraphedRegistrationStep.__init__)


class GraphedRegistrationStep:
    """The s2 inner iteration (registration_step above: bind -> render -> L1/SSIM [+ hinges] -> backward ->
    densification stats -> Adam) captured ONCE into a hipGraph and replayed per iteration.

    The reference's loop is launch- and sync-bound on a fast GPU: ~60 small kernels and several host round trips
    per iteration for ~1 ms of GPU work.  A replay is one graph launch behind ONE 184-byte H2D copy (camera matrices,
    tangents and the device addresses of this iteration's ground truth / mask / mask tile table, which the loss kernels read through a
    pointer table -- images already on the GPU are not copied) and in front of ONE 48-byte read-back (loss statistics +
    the rasterizer's overflow word); the learning rates live on the device (GraphAdam.push_lr).  The rasterizer
    runs with the binning capacity learnt during the eager warm-up; if a replayed step overflows it, the guarded
    kernels leave parameters, moments and statistics untouched, and this class grows the capacity, re-captures and
    replays the step -- so results never depend on the capacity guess.

    gaussians.optimizer must be a GraphAdam (ggsplat.adam).  Camera objects need world_view_transform,
    full_proj_transform, camera_center, FoVx, FoVy (scene/cameras.py attributes).

    lean=True (default) assembles the iteration directly from the C entry points -- mesh binding, rasterizer
    forward/backward, fused photometric loss, ggs_registration_aux (hinges, opacity chain rule, densification
    statistics), guarded Adam: ~30 kernels per iteration, no autograd, loss values returned as Python floats read back
    with the overflow flag.  lean=False captures registration_step() itself (autograd + ~120 small PyTorch kernels);
    both are tested against the eager step."""

    def __init__(self, gaussians, W: int, H: int, bg, opt=DEFAULT_OPT, pipe=DEFAULT_PIPE,
                 first_frame_template: bool = True, track_densification: bool = True, use_mask: bool = True,
                 capacity_slack: float = 1.0, lean: bool = True, sparse_mask: Optional[bool] = None):
        if not isinstance(gaussians.optimizer, GraphAdam):
            raise TypeError("GraphedRegistrationStep needs gaussians.optimizer to be a ggsplat.adam.GraphAdam")
        dev = gaussians._xyz.device
        self.g, self.opt, self.pipe, self.bg = gaussians, opt, pipe, bg
        self.fft, self.track = first_frame_template, track_densification
        # Everything that changes from one iteration to the next sits in ONE 184-byte device block, refreshed by one H2D copy
        # from a pinned staging block: view [16] | full projection [16] | camera centre [3] | tan(fov/2) [2] | pad [3] floats,
        # then three device pointers (ground truth, mask, the mask's tile occupancy) that the loss kernels read through
        # (ggs_photometric_*_tab / _sparse).
        self._blk = torch.zeros(_BLK_BYTES, dtype=torch.uint8, device=dev)
        self._blk_host = torch.zeros(_BLK_BYTES, dtype=torch.uint8).pin_memory()
        f = self._blk[:160].view(torch.float32)
        self.cam = SimpleNamespace(
            image_height=H, image_width=W, world_view_transform=f[0:16].view(4, 4), full_proj_transform=f[16:32].view(4, 4),
            camera_center=f[32:35], tanfov=f[35:37].view(1, 2))
        self._ptrs = self._blk[160:_BLK_BYTES].view(torch.int64)
        self._host_f = self._blk_host[:160].view(torch.float32)
        self._host_p = self._blk_host[160:_BLK_BYTES].view(torch.int64)
        # Sparse-mask form of the loss's first pass (ggs_photometric_forward_sparse; lean form only): True / False, or None =
        # decided by the first mask this object sees (a silhouette that leaves most 16x16 tiles empty -> True).  The choice is
        # part of the captured graph; a later mask of the other kind still gives the right numbers, only not the faster kernel.
        self._sparse = sparse_mask if (lean and use_mask and opt.only_foreground_loss) else False
        self._mask_tiles: Dict[int, tuple] = {}
        self._cam_cache: Dict[int, tuple] = {}
        self._keep = None                                   # the tensors the pointer table names, alive until the replay is done
        self.gt = torch.zeros(3, H, W, device=dev)          # landing buffers for images that arrive on the host
        self.mask = torch.ones(1, H, W, device=dev) if use_mask else None
        # what the host reads back per iteration, in one D2H copy: 8 floats of loss statistics + the bin header (2 x int64)
        self._out = torch.zeros(48, dtype=torch.uint8, device=dev)
        self._out_host = torch.zeros(48, dtype=torch.uint8).pin_memory()
        self._hdr_host = self._out_host[32:48].view(torch.int64)
        self.lean = bool(lean)
        lam = float(opt.lambda_dssim)
        self._w = torch.tensor([[1.0 - lam, -lam]], device=dev)          # d loss / d {mean|x-y|, mean ssim}
        self._stats = self._out[:32].view(torch.float32)                  # {sum|x-y|, sum ssim, loss_xyz, loss_scale, n_vis}
        self._stats_host = self._out_host[:32].view(torch.float32)
        self._aux_scratch = torch.zeros(4, device=dev)
        self._sums = torch.zeros(2, device=dev)                            # photometric sums of the step (lean form)
        # Lean form: the parameter block is READ and the result block WRITTEN in place in pinned host memory by the step's
        # first / last kernel (ggs_step_prologue, ggs_registration_aux_tail) when that memory is mapped into the device's
        # address space -- no copy launch on either side of the replay.
        from ._lib import host_mapped_pointer
        self._blk_map = host_mapped_pointer(self._blk_host) if self.lean else 0
        self._out_map = host_mapped_pointer(self._out_host) if self.lean else 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Dict[str, torch.Tensor] = {}
        self.recaptures = 0
        self.optimizer = gaussians.optimizer
        self._slack = float(capacity_slack)     # applied once to the learnt capacity (< 1 exercises the recovery path)

    def _packed_camera(self, cam) -> torch.Tensor:
        """The 37 floats of a camera (matrices, centre, tangents) as a host tensor, packed once per camera object (the
        reference's Camera keeps its matrices on the GPU, scene/cameras.py:59-62: one read-back each, at first use)."""
        import math
        mats = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
        vers = tuple(getattr(m, "_version", 0) for m in mats) + (float(cam.FoVx), float(cam.FoVy))
        ent = self._cam_cache.get(id(cam))
        if ent is not None and ent[0] is cam and ent[2] == vers and all(a is b for a, b in zip(ent[3], mats)):
            return ent[1]
        t = torch.zeros(40)
        t[0:16] = mats[0].detach().reshape(16).float().cpu()
        t[16:32] = mats[1].detach().reshape(16).float().cpu()
        t[32:35] = mats[2].detach().reshape(3).float().cpu()
        t[35], t[36] = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        if len(self._cam_cache) > 4096:
            self._cam_cache.clear()
        # the entry holds the camera and its tensors (no id() reuse) and their version counters (an in-place edit re-packs)
        self._cam_cache[id(cam)] = (cam, t, vers, mats)
        return t

    def _resident(self, t, landing, shape):
        """A contiguous fp32 device tensor holding `t`: `t` itself when it already is one (no copy), else the landing buffer."""
        if t.device == landing.device and t.dtype is torch.float32 and t.is_contiguous() and t.numel() == landing.numel():
            return t
        landing.copy_(t.reshape(shape), non_blocking=True)
        return landing

    def _tiles_of(self, m):
        """Tile occupancy of mask tensor `m` (ggs_mask_tiles), computed once per (tensor, version); the first one also settles
        sparse_mask=None: sparse iff fewer than half of the tiles hold a mask pixel (one read-back, once).  The cache holds
        the masks by WEAK reference: a caller that uploads a fresh mask every iteration (`gt_mask.cuda()`, as the reference's
        loop does with its images) must not find 8 MB per iteration kept alive here."""
        import weakref
        from .loss import mask_tile_occupancy
        cache = self._mask_tiles
        ent = cache.get(id(m))
        if ent is None or ent[0]() is not m or ent[1] != m._version:
            if len(cache) > 256:
                for k in [k for k, e in cache.items() if e[0]() is None]:
                    del cache[k]
                if len(cache) > 4096:
                    cache.clear()
            ent = cache[id(m)] = (weakref.ref(m), m._version, mask_tile_occupancy(m))
        if self._sparse is None:
            t = ent[2]
            self._sparse = bool(int((t != 0).sum()) * 2 < t.numel())
            if not self._sparse:
                return None
        return ent[2]

    def _load(self, cam, gt_image, mask):
        self._host_f[:40] = self._packed_camera(cam)
        gt = self._resident(gt_image, self.gt, self.gt.shape)
        m = None
        if self.mask is not None:
            m = self.mask if mask is None else self._resident(mask, self.mask, self.mask.shape)
        mt = self._tiles_of(m) if (m is not None and self._sparse is not False) else None
        self._keep = (gt, m, mt)
        self._host_p[0], self._host_p[1] = gt.data_ptr(), (m.data_ptr() if m is not None else 0)
        self._host_p[2] = mt.data_ptr() if mt is not None else 0
        if not self._blk_map:
            self._blk.copy_(self._blk_host, non_blocking=True)
        if not self.lean:           # the autograd form reads the images from the static buffers
            if gt is not self.gt:
                self.gt.copy_(gt, non_blocking=True)
            if m is not None and m is not self.mask:
                self.mask.copy_(m, non_blocking=True)

    def _body_lean(self, optimizer_step: bool, track: bool):
        """registration_step() without autograd: the same arithmetic in the same order, called through the C ABI.  14 launches:
        prologue (zero fills | parameter block | sigmoid | mesh binding) -> forward (5) -> loss (2) -> backward (2) -> mesh
        binding backward -> regularisers / statistics / result block (2) -> Adam."""
        import ctypes as C
        from ._lib import GgsStepPrologue, GgsStepTail, check, lib, ptr
        g, opt, cam = self.g, self.opt, self.cam
        L = lib()
        dev = g._xyz.device
        stream = stream_ptr(dev)
        H, W = cam.image_height, cam.image_width
        P, Fn = g._xyz.shape[0], g.mesh.f.shape[0]
        try:
            return self._lean_calls(L, g, opt, cam, dev, stream, H, W, P, Fn, optimizer_step, track)
        finally:
            L.ggs_step_end()            # a step that raised half way must not leave pre-clear marks behind (include/ggsplat.h)

    def _lean_calls(self, L, g, opt, cam, dev, stream, H, W, P, Fn, optimizer_step: bool, track: bool):
        import ctypes as C
        from ._lib import GgsStepPrologue, GgsStepTail, check, ptr
        with torch.no_grad():
            verts, faces, binding, bary = g.mesh.v, g.mesh.f, g.binding, g.gs_bc
            xyz, scaling, rot = torch.empty_like(g._xyz), torch.empty_like(g._scaling), torch.empty_like(g._rotation)
            opacity = torch.empty_like(g._opacity)
            d_verts = torch.empty_like(verts)
            K = 1 + g._features_rest.shape[1]
            ws = R.plan_step(P, K, g.active_sh_degree, W, H, 1, dev)
            pro = GgsStepPrologue()
            # (range, bytes, does a later call of this step zero-fill it itself?) -- only those are MARKED for their consumer:
            # ggs_forward (binning counters), ggs_backward (gradient records), ggs_photometric_forward_roi (sums) and, when the
            # hinge terms run, ggs_registration_aux_tail (its 16-byte sums); nothing in the library clears dL/dvertices
            clears = ((ws.bin, ws.bin_clear, True), (ws.scratch, ws.scratch_clear, True), (self._sums, 8, True),
                      (self._aux_scratch, 16, bool(self.fft)), (d_verts, d_verts.numel() * 4, False))
            pro.n_clear = len(clears)
            for i, (t, n, consumed) in enumerate(clears):
                pro.clear_ptr[i], pro.clear_bytes[i] = t.data_ptr(), n
                pro.consumer_mask |= int(consumed) << i
            if self._blk_map:
                pro.copy_src, pro.copy_dst, pro.copy_bytes = self._blk_map, self._blk.data_ptr(), _BLK_BYTES
            pro.P, pro.F = P, Fn
            pro.verts, pro.faces, pro.binding = ptr(verts), ptr(faces), ptr(binding)
            pro.local_xyz, pro.log_scaling, pro.raw_rot, pro.bary = ptr(g._xyz), ptr(g._scaling), ptr(g._rotation), ptr(bary)
            pro.xyz, pro.scaling, pro.rotation = ptr(xyz), ptr(scaling), ptr(rot)
            pro.n_opacity, pro.opacity_logit, pro.opacity = P, ptr(g._opacity), ptr(opacity)
            check(L.ggs_step_prologue(C.byref(pro), stream), "ggs_step_prologue")
            shs = g._features_dc if K == 1 else torch.cat((g._features_dc, g._features_rest), dim=1)
            color, radii, _, _, st = R.forward_views(
                xyz, opacity, shs, None, scaling, rot, None, view=cam.world_view_transform,
                proj=cam.full_proj_transform, campos=cam.camera_center, tanfov=cam.tanfov, bg=self.bg, W=W, H=H,
                sh_degree=g.active_sh_degree, debug=self.pipe.debug, workspaces=ws)
            hdr = R.last_header()
            use_m = self.mask is not None and opt.only_foreground_loss
            gt_tab, m_tab = self._ptrs[0:1], (self._ptrs[1:2] if use_m else None)
            scratch = torch.empty(L.ggs_photometric_scratch_bytes(1, H, W), device=dev, dtype=torch.uint8)
            # region-of-interest form: dL/dimage only where the backward below reads it (tiles with a list: ~1 in 10 here)
            tc = R.last_tile_count()
            if use_m and self._sparse:       # silhouette mask: the boxes without a mask pixel are skipped
                check(L.ggs_photometric_forward_sparse(1, H, W, ptr(color), None, None, ptr(gt_tab), ptr(m_tab), ptr(tc), None,
                                                       ptr(self._ptrs[2:3]), ptr(self._sums), ptr(scratch), stream),
                      "ggs_photometric_forward_sparse")
            else:
                check(L.ggs_photometric_forward_roi(1, H, W, ptr(color), None, None, ptr(gt_tab), ptr(m_tab), ptr(tc),
                                                    ptr(self._sums), ptr(scratch), stream), "ggs_photometric_forward_roi")
            dimg = torch.empty_like(color)
            check(L.ggs_photometric_backward_roi(1, H, W, ptr(color), None, None, ptr(gt_tab), ptr(m_tab), ptr(tc), ptr(scratch),
                                                 ptr(self._w), ptr(dimg), stream), "ggs_photometric_backward_roi")
            gr = R.backward_views(st, dimg, want_means2D=True, scratch=ws.scratch)
            d_xyz, d_ls, d_rr = torch.empty_like(g._xyz), torch.empty_like(g._scaling), torch.empty_like(g._rotation)
            check(L.ggs_mesh_bind_backward(P, Fn, ptr(verts), ptr(faces), ptr(binding), ptr(g._xyz), ptr(g._scaling),
                                           ptr(g._rotation), ptr(bary), ptr(gr["means3D"]), ptr(gr["scales"]),
                                           ptr(gr["rotations"]), ptr(d_verts), ptr(d_xyz), ptr(d_ls), ptr(d_rr), stream),
                  "ggs_mesh_bind_backward")
            d_op = torch.empty_like(g._opacity)
            stats = self.fft and track
            # the result block {photometric sums, hinge losses, n_visible | bin header}: written by the last regulariser
            # kernel, straight into the pinned host block when that is mapped (else into self._out, copied back by __call__)
            tail = GgsStepTail(ptr(self._sums), ptr(hdr), self._out_map or self._out.data_ptr())
            step_now = optimizer_step and g.optimizer is not None
            if step_now:                    # ... and advances the optimiser states, so that the update is ONE launch
                states = g.optimizer.tick_states()
                tail.n_adam_states = len(states)
                for i, a in enumerate(states):
                    tail.adam_states[i] = a
                tail.beta1, tail.beta2 = g.optimizer.betas
            check(L.ggs_registration_aux_tail(
                P, ptr(g._xyz), ptr(g._scaling), ptr(radii), ptr(gr["means2D"]), ptr(opacity), ptr(gr["opacities"]),
                ptr(d_op), float(opt.threshold_xyz), float(opt.lambda_xyz), float(opt.threshold_scale),
                float(opt.lambda_scale), ptr(d_xyz) if self.fft else None, ptr(d_ls) if self.fft else None,
                ptr(g.max_radii2D) if stats else None, ptr(g.xyz_gradient_accum) if stats else None,
                ptr(g.denom) if stats else None, None, ptr(self._aux_scratch), ptr(hdr[1:2]), C.byref(tail), stream),
                "ggs_registration_aux_tail")
            if step_now:
                g.mesh.v.grad, g._xyz.grad, g._scaling.grad, g._rotation.grad, g._opacity.grad = d_verts, d_xyz, d_ls, d_rr, d_op
                gs = gr["shs"]
                g._features_dc.grad = gs if K == 1 else gs[:, :1].contiguous()
                g._features_rest.grad = torch.empty_like(g._features_rest) if K == 1 else gs[:, 1:].contiguous()
                g.optimizer.step(guard=hdr[1:2], tick=False)
                g.optimizer.zero_grad()
        return {}

    def _losses_from_stats(self) -> Dict[str, float]:
        s, opt = self._stats_host, self.opt
        n = 3.0 * self.cam.image_height * self.cam.image_width
        lam = float(opt.lambda_dssim)
        d = {"img": float(s[0]) / n * (1.0 - lam), "ssim": 1.0 - float(s[1]) / n * lam}
        if self.fft:
            d["xyz"], d["scale"], d["n_visible"] = float(s[2]), float(s[3]), float(s[4])
        d["loss"] = sum(v for k, v in d.items() if k != "n_visible")
        return d

    def _body(self, optimizer_step: bool, track: bool):
        if self.lean:
            return self._body_lean(optimizer_step, track)
        d = registration_step(self.g, self.cam, self.gt, self.mask, self.bg, self.opt, self.pipe,
                              first_frame_template=self.fft, track_densification=track,
                              optimizer_step=optimizer_step, fused_loss=True)
        d.pop("render_pkg", None)
        return d

    def _capture(self):
        g = self.g
        # eager warm-up on a side stream: learns the binning capacity, leaves parameters and statistics alone
        import warnings
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), warnings.catch_warnings():
            # the warm-up runs on a side stream as torch.cuda.graph's documentation prescribes; leaves created on the
            # default stream make autograd note the stream change -- expected here, and the grads are dropped right after
            warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
            self._body(optimizer_step=False, track=False)
            self.optimizer.zero_grad()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self._slack != 1.0:
            R.grow_capacity(self._slack)
            self._slack = 1.0
        from . import profile
        profile.restart_if_active()          # a timestamp profile (ggsplat.profile) logs the captured launches only, not the warm-up's
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")   # capture stream
            out = self._body(optimizer_step=True, track=self.track)
            self._hdr_dev = R.last_header()
            if not self.lean:                    # (the lean form's last regulariser kernel assembles the whole block)
                torch.add(self._hdr_dev, 0, out=self._out[32:48].view(torch.int64))   # next to the statistics: one read-back
        self.out = {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}
        self._captured = self._identity()

    def _identity(self):
        """The tensors whose ADDRESSES a captured step holds: parameters, mesh connectivity, binding, statistics and the
        optimiser's per-parameter state (moments, step words).  Density control (ggsplat.densify) and load_ply replace them
        with new tensors; the next call re-captures.  The list holds STRONG REFERENCES and is compared with `is`
        (_same_tensors): an (id, data_ptr, shape) fingerprint can be reproduced by the caching allocator and CPython handing
        the addresses of dead tensors to new ones -- a density step that selects nothing keeps every shape -- while the
        moments land somewhere else (MeshGaussianModel._bind documents the same pitfall)."""
        g = self.g
        ts = list(g.parameters()) + [g.mesh.f, g.binding, g.gs_bc, g.max_radii2D, g.xyz_gradient_accum, g.denom]
        opt = getattr(self, "optimizer", None) or g.optimizer
        for group in (opt.param_groups if opt is not None else ()):
            for p in group["params"]:
                st = opt.state.get(p)
                ts.append(p)
                if st:
                    ts.extend(v for v in st.values() if torch.is_tensor(v))
        return ts

    @staticmethod
    def _same_tensors(a, b) -> bool:
        return a is not None and len(a) == len(b) and all(x is y for x, y in zip(a, b))

    def _ready_graph(self):
        if self.graph is not None and not self._same_tensors(self._captured, self._identity()):
            self.graph = None                       # P changed under the graph (densify / prune): capture the new shapes
            self.recaptures += 1
        if self.graph is None:
            self._capture()
            # the capture itself does not execute anything: the caller replays

    # ---- split form of __call__ for PipelinedRegistrationStep: queue an iteration / read its result later -----------------
    def launch(self, cam, gt_image, mask=None) -> None:
        """Queue one iteration on the current stream WITHOUT waiting for it: parameter block, graph replay, result read-back
        (in place when the result block is mapped pinned memory), an event behind them.  collect() reads the result."""
        self._last_args = (cam, gt_image, mask)
        self._load(cam, gt_image, mask)
        self._ready_graph()
        self.graph.replay()
        if not self._out_map:
            self._out_host.copy_(self._out, non_blocking=True)
        if getattr(self, "_done", None) is None:
            self._done = torch.cuda.Event()
        self._done.record()

    def collect(self):
        """Wait for the iteration queued by launch().  Returns its loss dictionary, or None when its forward overflowed the
        static binning capacity (nothing was updated: guarded kernels) -- the caller re-runs it."""
        self._done.synchronize()
        self._keep = None
        if int(self._hdr_host[1]) != 0:
            return None
        return self._losses_from_stats() if self.lean else self.out

    def __call__(self, cam, gt_image, mask=None) -> Dict[str, torch.Tensor]:
        """One optimisation step.  Returns {"loss", "img", "ssim", ...}: Python floats (lean) or device scalars that the
        next call overwrites (lean=False)."""
        self._load(cam, gt_image, mask)
        self._ready_graph()
        while True:
            self.graph.replay()
            if not self._out_map:
                self._out_host.copy_(self._out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            self._keep = None
            if int(self._hdr_host[1]) == 0:
                return self._losses_from_stats() if self.lean else self.out
            # the static binning capacity was too small for this view: nothing was updated (guarded kernels)
            R.grow_capacity(2.0)
            self.recaptures += 1
            self.graph = None
            self._capture()


class PipelinedRegistrationStep:
    """Two captured copies of the lean s2 iteration (GraphedRegistrationStep) replayed ALTERNATELY on one stream, the result of
    iteration i read while iteration i + 1 runs.

    A replayed iteration is ~0.35 ms of kernels; a caller that waits for each result before it queues the next iteration leaves
    the GPU idle for the host's share of the period -- writing the 184-byte parameter block, hipGraphLaunch, waking up from the
    wait, reading the 48-byte result (~20-35 us per iteration).  Here iteration i + 1 is queued BEFORE the host
    waits for iteration i: the stream runs the two graphs back to back (parameters, optimiser state and statistics are shared:
    same addresses in both captures, so iteration i + 1 sees the update of iteration i), and every call returns the losses of the
    PREVIOUS call (None for the first; flush() returns the last).  Two captures because each reads its per-iteration block
    (camera, image pointers) from its own pinned host block in place: the host may fill block B while the GPU still reads A.

    The reference's loop uses the loss only for its progress bar (s2_registration.py:284-292), so the one-iteration lag costs
    nothing there.  What differs from the sequential form: an iteration whose forward overflowed the static binning capacity
    (it changed nothing: guarded kernels) is re-run AFTER the iteration that was already queued behind it -- the two updates
    swap places.  That happens at most a few times per run (the capacity doubles each time).
    Density control between two calls needs no flush(): it runs on the same stream behind the queued iteration, and each copy
    re-captures itself on its next launch when the tensors it captured were replaced (GraphedRegistrationStep._ready_graph)."""

    def __init__(self, gaussians, W: int, H: int, bg, **kw):
        kw["lean"] = True
        self.steps = [GraphedRegistrationStep(gaussians, W, H, bg, **kw) for _ in range(2)]
        self.steps[1]._mask_tiles = self.steps[0]._mask_tiles          # one tile table per mask, whichever copy meets it first
        self._pending: Optional[GraphedRegistrationStep] = None
        self._n = 0

    @property
    def recaptures(self) -> int:
        return sum(s.recaptures for s in self.steps)

    def _recover(self, first: GraphedRegistrationStep, second: Optional[GraphedRegistrationStep]) -> Dict[str, float]:
        """`first` overflowed.  Drain the stream, grow the capacity, drop both captures, re-run `first` (and `second` if it
        overflowed too) through the sequential form, which re-captures and retries until the capacity fits."""
        second_ok = None if second is None else second.collect()
        torch.cuda.synchronize()
        R.grow_capacity(2.0)
        for s in self.steps:
            s.graph = None
            s.recaptures += 1
        out = first(*first._last_args)
        if second is not None and second_ok is None:
            second_ok = second(*second._last_args)
        self._late = second_ok           # the result of the iteration queued behind the overflowed one
        return out

    def __call__(self, cam, gt_image, mask=None) -> Optional[Dict[str, float]]:
        """Queue iteration n; return the losses of iteration n - 1 (None for n = 0)."""
        cur = self.steps[self._n & 1]
        self._n += 1
        ready, self._ready = getattr(self, "_ready", None), None
        prev, self._pending = self._pending, cur
        cur.launch(cam, gt_image, mask)
        for s in self.steps:             # sparse_mask=None is settled ONCE, by the first mask either copy meets: both captures
            if s._sparse is None and cur._sparse is not None:          # then run the same loss kernels
                s._sparse = cur._sparse
        if prev is None:
            return ready                 # (a result the recovery below already collected, or None for the first call)
        out = prev.collect()
        if out is None:                  # iteration n - 1 overflowed: recover both, nothing is pending afterwards
            out = self._recover(prev, cur)
            self._pending, self._ready = None, self._late
        return out

    def flush(self) -> Optional[Dict[str, float]]:
        """Wait for the last queued iteration and return its losses (None if there is none)."""
        cur, self._pending = self._pending, None
        if cur is None:
            out, self._ready = getattr(self, "_ready", None), None
            return out
        out = cur.collect()
        if out is None:
            out = self._recover(cur, None)
        return out


class ReplicaRegistrationSteps:
    """R INDEPENDENT registrations on one GPU, each with the reference's own semantics -- one optimiser step per rendered view
    (s2_registration.py:241-251, :326) -- replayed side by side: replica r owns a model, an optimiser, its workspaces, its captured
    iteration (GraphedRegistrationStep) and a stream of its own.

    Strict reference semantics cannot be spread over views (iteration i + 1 reads the parameters iteration i wrote: SURVEY 8e
    "replicas only"), and ONE such iteration leaves most of an MI355X idle: a single 1080p view is ~1 100 non-empty tiles = 4 400
    quadrant waves for 1 024 SIMDs, and the kernels are as long as their dozen longest walks.  Independent
    problems -- the frames of several sequences, several subjects, a hyper-parameter sweep -- fill that idle time: every call queues
    one iteration of every replica on the replica's stream (one hipGraphLaunch each, ~12 us of host time) and then collects the R
    results; the GPU interleaves the R graphs.  On 8 GPUs the same thing runs as 8 x R replicas with no collective at all.

    Each replica's arithmetic is exactly that of its solo GraphedRegistrationStep: same kernels on the same inputs; results differ
    from a solo run only by the order of the float atomics inside the render backward, as two solo runs differ from each other
    (tests/test_gpu_graph_step.py::test_replicas_follow_their_solo_trajectories)."""

    def __init__(self, models, W: int, H: int, bg, **kw):
        kw["lean"] = True
        self.models = list(models)
        dev = self.models[0]._xyz.device
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.models]
        self.steps: List[GraphedRegistrationStep] = []
        for m, s in zip(self.models, self.streams):
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                self.steps.append(GraphedRegistrationStep(m, W, H, bg, **kw))

    def __len__(self) -> int:
        return len(self.steps)

    @property
    def recaptures(self) -> int:
        return sum(s.recaptures for s in self.steps)

    def __call__(self, cams, gt_images, masks=None) -> List[Dict[str, float]]:
        """One iteration of EVERY replica: cams[r], gt_images[r] (and masks[r]) go to replica r.  Returns the R loss dictionaries."""
        R_ = len(self.steps)
        masks = [None] * R_ if masks is None else masks
        for st, s, c, g, m in zip(self.steps, self.streams, cams, gt_images, masks):
            with torch.cuda.stream(s):
                st.launch(c, g, m)
        outs = []
        for st, s in zip(self.steps, self.streams):
            out = st.collect()
            if out is None:              # this replica's forward overflowed its static capacity (nothing was updated): the sequential
                with torch.cuda.stream(s):                          # form grows the capacity, re-captures and repeats the iteration
                    out = st(*st._last_args)
            outs.append(out)
        return outs

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()


class GraphedAppearanceStep(GraphedRegistrationStep):
    """The s3 inner iteration (appearance_step above: net -> offsets -> render(vis_mask) -> five-term loss -> backward ->
    Adam over the net's and the Gaussians' parameters) captured into a hipGraph and replayed per iteration.

    `net(gaussians, cam)` must be capturable: static shapes, no host syncs (the StyleUNet on PyTorch-ROCm is; an open3d
    ray cast is not -- `ggsplat.mesh_gaussian_model.visible_mask` is the GPU replacement).  The visibility mask is applied
    to the opacities (`pipe.mask_by_opacity`) instead of gathering the visible Gaussians: same image, same gradients,
    static shapes.  `optimizer` is a GraphAdam over whatever the step trains."""

    def __init__(self, gaussians, net: Callable, W: int, H: int, bg, optimizer: GraphAdam, opt=DEFAULT_OPT,
                 pipe=DEFAULT_PIPE, use_mask: bool = True, capacity_slack: float = 1.0):
        if not isinstance(optimizer, GraphAdam):
            raise TypeError("GraphedAppearanceStep needs a ggsplat.adam.GraphAdam")
        saved = gaussians.optimizer
        gaussians.optimizer = optimizer                      # the base constructor checks / records gaussians.optimizer
        try:
            super().__init__(gaussians, W, H, bg, opt=opt, pipe=SimpleNamespace(**{**vars(pipe), "mask_by_opacity": True}),
                             first_frame_template=False, track_densification=False, use_mask=use_mask,
                             capacity_slack=capacity_slack, lean=False)
        finally:
            gaussians.optimizer = saved
        self.net = net
        self.optimizer = optimizer

    def _body(self, optimizer_step: bool, track: bool):
        d = appearance_step(self.g, self.net, self.cam, self.gt, self.mask, self.bg,
                            optimizer=self.optimizer if optimizer_step else None, opt=self.opt, pipe=self.pipe,
                            fused_loss=True)
        d.pop("render_pkg", None)
        if not optimizer_step:
            self.optimizer.zero_grad()
        return d
