"""Adaptive density control of the mesh-bound Gaussians: clone / split / prune with `binding` inheritance and
optimiser-state surgery -- the host-side counterpart of scene/mesh_gaussian_model.py:130-208 (prune_points,
densify_and_split, densify_and_clone) and scene/gaussian_model.py:276-408 (_prune_optimizer, cat_tensors_to_optimizer,
densification_postfix, densify_and_prune), as the first-frame loop calls them (s2_registration.py:310-322).

Same arithmetic and the same order of operations as the reference, including its quirks:
  * a face never loses its last Gaussian: a prune request that would empty a face is dropped for ALL the Gaussians of that
    face in the request (scene/mesh_gaussian_model.py:131-137);
  * a split sample is  R(_rotation) n + get_xyz  with n ~ N(0, get_scaling), i.e. a WORLD-frame position written into the
    local `_xyz` of the children, whose log-scale is  log(get_scaling / face_scaling / (0.8 N))
    (scene/mesh_gaussian_model.py:168-177);
  * the statistics (xyz_gradient_accum, denom, max_radii2D) are zeroed for EVERY Gaussian after a clone / split
    (scene/gaussian_model.py:350-352).
Works with torch.optim.Adam and with ggsplat.adam.GraphAdam (whose per-parameter step words travel with the moments).
Every call changes P: parameters become NEW tensors, so a captured step (GraphedRegistrationStep) re-captures itself the
next time it is called (it keeps the identities of the tensors it captured).

These are mix-in methods of ggsplat.mesh_gaussian_model.MeshGaussianModel.  PyTorch indexing / cat only: the step runs
once every `densification_interval` (100) iterations, it is not on the hot path.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

_GROUP_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
               "scaling": "_scaling", "rotation": "_rotation"}


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:91-113: rotation matrices of (w, x, y, z) quaternions, normalised first."""
    q = r / torch.sqrt((r * r).sum(1))[:, None]
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)


class DensifyMixin:
    # ---- optimiser-state surgery (scene/gaussian_model.py:276-333) -------------------------------------------------
    def _swap_param(self, group: Dict, new: torch.Tensor, fix_state) -> nn.Parameter:
        """Replace the single parameter of `group` by nn.Parameter(new); its optimiser state (if any) moves to the new
        key after `fix_state(state)` resized the moments."""
        opt = self.optimizer
        old = group["params"][0]
        stored = opt.state.get(old, None)
        p = nn.Parameter(new.contiguous().requires_grad_(True))
        if stored is not None:
            fix_state(stored)
            del opt.state[old]
            opt.state[p] = stored
        group["params"][0] = p
        return p

    def _prune_optimizer(self, mask: torch.Tensor) -> Dict[str, nn.Parameter]:
        out = {}
        for group in self.optimizer.param_groups:
            if group.get("name") == "vertex":
                continue

            def fix(st):
                st["exp_avg"] = st["exp_avg"][mask].contiguous()
                st["exp_avg_sq"] = st["exp_avg_sq"][mask].contiguous()
            out[group["name"]] = self._swap_param(group, group["params"][0].detach()[mask], fix)
        return out

    def cat_tensors_to_optimizer(self, tensors_dict: Dict[str, torch.Tensor]) -> Dict[str, nn.Parameter]:
        out = {}
        for group in self.optimizer.param_groups:
            if group.get("name") == "vertex":
                continue
            assert len(group["params"]) == 1
            ext = tensors_dict[group["name"]]

            def fix(st, ext=ext):
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            out[group["name"]] = self._swap_param(group, torch.cat((group["params"][0].detach(), ext), dim=0), fix)
        return out

    def _adopt(self, tensors: Dict[str, nn.Parameter]) -> None:
        for name, attr in _GROUP_ATTR.items():
            setattr(self, attr, tensors[name])
        self._bound = None

    def _ensure_binding_counter(self) -> None:
        bc = getattr(self, "binding_counter", None)
        Fn = self.mesh.f.shape[0]
        if bc is None or bc.shape[0] != Fn:
            self.binding_counter = torch.bincount(self.binding, minlength=Fn).to(torch.int32)

    # ---- scene/mesh_gaussian_model.py:130-156 -----------------------------------------------------------------------
    @torch.no_grad()
    def prune_points(self, mask: torch.Tensor) -> None:
        """Remove the Gaussians where `mask` is True -- except that no face loses all its Gaussians."""
        self._ensure_binding_counter()
        mask = mask.clone()
        binding_to_prune = self.binding[mask]
        counter_prune = torch.zeros_like(self.binding_counter)
        counter_prune.scatter_add_(0, binding_to_prune, torch.ones_like(binding_to_prune, dtype=torch.int32))
        mask_redundant = (self.binding_counter - counter_prune) > 0
        mask[mask.clone()] = mask_redundant[binding_to_prune]

        valid = ~mask
        self._adopt(self._prune_optimizer(valid))
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]
        gone = self.binding[mask]
        self.binding_counter.scatter_add_(0, gone, -torch.ones_like(gone, dtype=torch.int32))
        self.binding = self.binding[valid].contiguous()
        if self.gs_bc is not None:
            self.gs_bc = self.gs_bc[valid].contiguous()

    # ---- scene/gaussian_model.py:335-352 ----------------------------------------------------------------------------
    @torch.no_grad()
    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation}
        self._adopt(self.cat_tensors_to_optimizer(d))
        n, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def _inherit_binding(self, new_binding: torch.Tensor, new_bc=None) -> None:
        self._ensure_binding_counter()
        self.binding = torch.cat((self.binding, new_binding)).contiguous()
        self.binding_counter.scatter_add_(0, new_binding, torch.ones_like(new_binding, dtype=torch.int32))
        if self.gs_bc is not None:
            self.gs_bc = torch.cat((self.gs_bc, new_bc)).contiguous()

    # ---- scene/mesh_gaussian_model.py:158-189 -----------------------------------------------------------------------
    @torch.no_grad()
    def densify_and_split(self, grads, grad_threshold, scene_extent, N: int = 2) -> None:
        n_init = self._xyz.shape[0]
        dev = self._xyz.device
        padded_grad = torch.zeros(n_init, device=dev)
        padded_grad[:grads.shape[0]] = grads.squeeze()
        sel = padded_grad >= grad_threshold
        scaling, xyz = self.get_scaling.detach(), self.get_xyz.detach()
        sel = torch.logical_and(sel, torch.max(scaling, dim=1).values > self.percent_dense * scene_extent)

        stds = scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(self._rotation.detach()[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz[sel].repeat(N, 1)
        face_scaling = self.face_scaling.detach()[self.binding[sel]]
        new_scaling = torch.log((scaling[sel] / face_scaling).repeat(N, 1) / (0.8 * N))
        new_rotation = self._rotation.detach()[sel].repeat(N, 1)
        new_features_dc = self._features_dc.detach()[sel].repeat(N, 1, 1)
        new_features_rest = self._features_rest.detach()[sel].repeat(N, 1, 1)
        new_opacity = self._opacity.detach()[sel].repeat(N, 1)
        self._inherit_binding(self.binding[sel].repeat(N), None if self.gs_bc is None else self.gs_bc[sel].repeat(N, 1))
        self.densification_postfix(new_xyz, new_features_dc, new_features_rest, new_opacity, new_scaling, new_rotation)
        prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool)))
        self.prune_points(prune_filter)

    # ---- scene/mesh_gaussian_model.py:191-208 -----------------------------------------------------------------------
    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent) -> None:
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel = torch.logical_and(sel, torch.max(self.get_scaling.detach(), dim=1).values <= self.percent_dense * scene_extent)
        new = [t.detach()[sel] for t in (self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling,
                                         self._rotation)]
        self._inherit_binding(self.binding[sel], None if self.gs_bc is None else self.gs_bc[sel])
        self.densification_postfix(*new)

    # ---- scene/gaussian_model.py:394-408 ----------------------------------------------------------------------------
    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size) -> None:
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune_mask = (self.get_opacity.detach() < min_opacity).squeeze()
        if max_screen_size:
            big_points_vs = self.max_radii2D > max_screen_size
            big_points_ws = self.get_scaling.detach().max(dim=1).values > 0.1 * extent
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
        self.prune_points(prune_mask)
