"""MeshGaussianModel -- host-side mirror of the reference's mesh-bound Gaussian model.

Mirrors the attribute surface ``render()`` and the s2 / s3 inner loops touch
(scene/gaussian_model.py:26-119, scene/mesh_gaussian_model.py:48-128, :350-379,
scene/avatar_gaussian_model.py:140-159):
  _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity, mesh.v, mesh.f, binding,
  active_sh_degree, max_sh_degree, get_xyz / get_scaling / get_rotation / get_opacity / get_features,
  get_final_xyz (+ local_xyz, shs, gs_bc for the s3 variant), update_face_coor(),
  training_setup(opt, is_ff), optimizer, add_densification_stats(), max_radii2D.

What differs: the reference evaluates update_face_coor + the three getters as ~25 small
PyTorch kernels (gathers, bmm, roma quaternion algebra) per optimisation step; here ONE fused
HIP kernel (ggs_mesh_bind_forward / _backward in libggsplat.so) produces xyz, scaling and
rotation together, and its backward scatters straight into mesh.v.grad.  The result is cached
per update_face_coor() call, so the three getters cost nothing extra.

Persistence: save_ply / load_ply in the reference's PLY layout incl. binding.pkl, OBJ mesh IO
(ggsplat.mesh_io).  Adaptive density control (densify_and_prune, prune_points, densify_and_split / _clone with binding
inheritance and optimiser-state surgery, scene/mesh_gaussian_model.py:130-208) lives in ggsplat.densify.  LBS is out of
scope; the model is built from tensors (``from_tensors``) or from a stage-2 directory (load_mesh + load_ply).
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace

import torch
from torch import nn

from ._lib import check, lib, ptr, stream_ptr
from .densify import DensifyMixin


def _stream(dev):
    return stream_ptr(dev)


def _f32c(t):
    """float32 + contiguous; the common case (already both) costs two attribute reads instead of two dispatcher calls."""
    if t is None or (t.dtype is torch.float32 and t.is_contiguous()):
        return t
    return t.contiguous().float()


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


class _MeshBind(torch.autograd.Function):
    """(verts, local_xyz, log_scaling, raw_rot) -> (xyz, scaling, rotation); faces / binding / bary constant."""

    @staticmethod
    def forward(ctx, verts, local_xyz, log_scaling, raw_rot, faces, binding, bary):
        if verts.device.type != "cuda":
            raise RuntimeError("ggsplat mesh binding runs on the GPU only (no CPU path in the product)")
        verts, local_xyz, log_scaling, raw_rot = _f32c(verts), _f32c(local_xyz), _f32c(log_scaling), _f32c(raw_rot)
        P, Fn = local_xyz.shape[0], faces.shape[0]
        xyz, scaling = torch.empty_like(local_xyz), torch.empty_like(log_scaling)
        rotation = torch.empty_like(raw_rot)
        check(lib().ggs_mesh_bind_forward(P, Fn, ptr(verts), ptr(faces), ptr(binding), ptr(local_xyz),
                                          ptr(log_scaling), ptr(raw_rot), ptr(bary), ptr(xyz), ptr(scaling),
                                          ptr(rotation), _stream(verts.device)), "ggs_mesh_bind_forward")
        ctx.save_for_backward(verts, local_xyz, log_scaling, raw_rot, faces, binding, bary)
        return xyz, scaling, rotation

    @staticmethod
    def backward(ctx, g_xyz, g_scaling, g_rot):
        verts, local_xyz, log_scaling, raw_rot, faces, binding, bary = ctx.saved_tensors
        P, Fn = local_xyz.shape[0], faces.shape[0]
        c = _f32c
        d_verts = torch.zeros_like(verts)
        d_local, d_ls, d_rr = torch.empty_like(local_xyz), torch.empty_like(log_scaling), torch.empty_like(raw_rot)
        check(lib().ggs_mesh_bind_backward(P, Fn, ptr(verts), ptr(faces), ptr(binding), ptr(local_xyz),
                                           ptr(log_scaling), ptr(raw_rot), ptr(bary), ptr(c(g_xyz)), ptr(c(g_scaling)),
                                           ptr(c(g_rot)), ptr(d_verts), ptr(d_local), ptr(d_ls), ptr(d_rr),
                                           _stream(verts.device)), "ggs_mesh_bind_backward")
        return d_verts, d_local, d_ls, d_rr, None, None, None


def mesh_bind(verts, faces, binding, local_xyz, log_scaling, raw_rot, bary=None):
    """Fused forward (differentiable).  faces [F,3] int64, binding [P] int64, bary [P,3] or None."""
    return _MeshBind.apply(verts, local_xyz, log_scaling, raw_rot, _c(faces), _c(binding), _f32c(bary))


def visible_mask(verts, faces, binding, targets, camera, return_first_hit: bool = False):
    """First-hit ray cast camera -> targets against the mesh on the GPU (ggs_visibility)."""
    if verts.device.type != "cuda":
        raise RuntimeError("ggsplat visibility runs on the GPU only (no CPU path in the product)")
    L = lib()
    dev = verts.device
    v, f = verts.detach().float().contiguous(), faces.long().contiguous()
    tg, bd = targets.detach().float().contiguous(), binding.long().contiguous()
    cam = camera.detach().float().reshape(3).to(dev).contiguous()
    P, Fn, Vn = tg.shape[0], f.shape[0], v.shape[0]
    cap = 32 * Fn + 65536
    scratch = torch.empty(L.ggs_visibility_scratch_bytes(Fn, Vn, cap), device=dev, dtype=torch.uint8)
    mask = torch.empty(P, device=dev, dtype=torch.uint8)
    first = torch.empty(P, device=dev, dtype=torch.int32) if return_first_hit else None
    check(L.ggs_visibility(P, Fn, Vn, ptr(v), ptr(f), ptr(cam), ptr(tg), ptr(bd), ptr(scratch), cap, ptr(mask), ptr(first),
                           _stream(dev)), "ggs_visibility")
    return (mask.bool(), first) if return_first_hit else mask.bool()


def _version_of(t):
    """Autograd version counter of a tensor (None for a missing input; inference-mode tensors have none and cannot be
    modified in place outside inference mode: a constant stands in)."""
    if t is None:
        return None
    try:
        return t._version
    except RuntimeError:
        return -1


class MeshGaussianModel(DensifyMixin):
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 1.0
        self.mesh = None
        self.binding = None
        self.gs_bc = None            # barycentric coords (a, b, c) -> AvatarGaussianModel origin
        self.local_xyz = None        # s3: _xyz + network offset (scene/avatar_net.py:82)
        self._bound = None           # cache of the fused op for the current mesh / parameters
        self.opacity_activation = torch.sigmoid
        self.scaling_activation = torch.exp
        self.rotation_activation = torch.nn.functional.normalize

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_tensors(cls, verts, faces, params, sh_degree: int, device="cuda", gs_bc=None) -> "MeshGaussianModel":
        m = cls(sh_degree)
        m.active_sh_degree = sh_degree
        m.mesh = SimpleNamespace(v=nn.Parameter(verts.to(device).float().contiguous()),
                                 f=faces.to(device).long().contiguous())
        for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            setattr(m, k, nn.Parameter(params[k].to(device).float().contiguous()))
        m.binding = params["binding"].to(device).long().contiguous()
        # Gaussians per face (scene/mesh_gaussian_model.py:83): pruning never empties a face
        m.binding_counter = torch.bincount(m.binding, minlength=m.mesh.f.shape[0]).to(torch.int32)
        m.max_radii2D = torch.zeros(m._xyz.shape[0], device=device)
        if gs_bc is not None:
            m.gs_bc = gs_bc.to(device).float().contiguous()      # [P,3]
        return m

    # ---- mesh binding -------------------------------------------------------------------------
    def update_face_coor(self):
        """Invalidate the cached binding; the fused kernel runs on the next getter
        (scene/mesh_gaussian_model.py:90-95 recomputes the face frames here)."""
        self._bound = None

    def _bind_key(self, final: bool):
        """What a cached binding is valid for.  The reference recomputes get_xyz / get_scaling / get_rotation from the
        current parameters on every call (scene/mesh_gaussian_model.py:105-128), so the cache must never outlive them:
        the key holds the identity AND the autograd version counter of every input (bumped by optimizer.step(), copy_,
        reset_opacity, load_ply ...) and the grad mode -- a binding first evaluated under torch.no_grad() carries no
        graph and must not be handed to a later render() that needs gradients."""
        local = self.local_xyz if final else self._xyz
        mesh = self.mesh
        ins = (mesh.v, local, self._scaling, self._rotation, self.gs_bc, mesh.f, self.binding)
        try:                    # (three calls per render(): the plain attribute read, not a function call per tensor)
            vers = tuple([None if t is None else t._version for t in ins])
        except RuntimeError:    # inference-mode tensors have no version counter
            vers = tuple(_version_of(t) for t in ins)
        return (final, torch.is_grad_enabled()), ins, vers

    def _bind(self, final: bool = False):
        mode, ins, vers = self._bind_key(final)
        b = self._bound
        # the cache entry keeps REFERENCES to its inputs and compares them with `is`: a freshly assigned tensor (the
        # per-frame LBS path replaces mesh.v) can reuse the id() of a dead one, a live reference cannot be reused
        if b is None or b[0] != mode or b[2] != vers or len(b[1]) != len(ins) or any(x is not y for x, y in zip(b[1], ins)):
            local = self.local_xyz if final else self._xyz
            self._bound = b = (mode, ins, vers, mesh_bind(self.mesh.v, self.mesh.f, self.binding, local, self._scaling,
                                                            self._rotation, self.gs_bc))
        return b[3]

    def _bound_final(self) -> bool:
        return self._bound is not None and self._bound[0][0] and self.local_xyz is not None

    @property
    def get_xyz(self):
        return self._bind()[0]

    @property
    def get_final_xyz(self):
        return self._bind(final=True)[0]

    @property
    def get_scaling(self):
        return self._bind(final=self._bound_final())[1]

    @property
    def get_rotation(self):
        return self._bind(final=self._bound_final())[2]

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        """Python cov3D path of render() (pipe.compute_cov3D_python): Sigma = L L^T with
        L = R(normalise(_rotation)) diag(modifier * get_scaling), stripped to (xx,xy,xz,yy,yz,zz) --
        scene/gaussian_model.py:27-31,118-119 + utils/general_utils.py:74-120.  Like the reference it uses the
        LOCAL rotation `_rotation`, not the mesh-bound one (SURVEY a11 caveat); s2 / s3 never enable this path."""
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)
        Lm = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = Lm @ Lm.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    # ---- face-level quantities of update_face_coor, derived from the same kernel ---------------
    def _face_probe(self):
        Fn = self.mesh.f.shape[0]
        dev = self.mesh.v.device
        z3 = torch.zeros(Fn, 3, device=dev)
        ident = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(Fn, 1)
        return mesh_bind(self.mesh.v, self.mesh.f, torch.arange(Fn, device=dev), z3, z3, ident)

    @property
    def face_center(self):
        return self._face_probe()[0]

    @property
    def face_scaling(self):
        return self._face_probe()[1][:, :1]

    @property
    def face_orien_quat(self):
        return self._face_probe()[2]

    @property
    def face_orien_mat(self):
        """[F,3,3], columns (a0, a1 = face normal, a2) -- scene/mesh_gaussian_model.py:90-95 publishes the matrix of
        utils/graphics_utils.py:118-137; here it is rebuilt from the kernel's unit quaternion (w,x,y,z)."""
        r, x, y, z = self.face_orien_quat.unbind(-1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)

    # ---- optimisation (scene/mesh_gaussian_model.py:350-379) ----------------------------------
    def training_setup(self, training_args, is_ff: bool, optimizer: str = "auto"):
        dev = self._xyz.device
        self.percent_dense = getattr(training_args, "percent_dense", 0.01)
        self.xyz_gradient_accum = torch.zeros((self._xyz.shape[0], 1), device=dev)
        self.denom = torch.zeros((self._xyz.shape[0], 1), device=dev)
        pos_lr = training_args.position_lr_init * self.spatial_lr_scale
        if is_ff:
            groups = [
                {"params": [self._xyz], "lr": pos_lr, "name": "xyz"},
                {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
                {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
                {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
                {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
                {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
                {"params": [self.mesh.v], "lr": pos_lr, "name": "vertex"},
            ]
        else:                                   # non-first frames optimise the mesh only
            groups = [{"params": [self.mesh.v], "lr": pos_lr, "name": "vertex"}]
        # The reference builds torch.optim.Adam(l, lr=0.0, eps=1e-15) (scene/mesh_gaussian_model.py:375).  On the GPU the same
        # update runs through ggsplat.adam.GraphAdam: same param_groups / state / step() / zero_grad() / state_dict() surface,
        # ONE launch for all seven tensors instead of ~8 foreach kernels per group (0.48 ms of host time per eager iteration,
        # 0.14 ms with torch's own fused=True: profiles/r04_host_profile.md), graph-capturable, results within 2e-6 of
        # torch.optim.Adam (tests/test_gpu_graph_step.py).  optimizer="torch" keeps the PyTorch class.
        if optimizer == "torch" or not self._xyz.is_cuda:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        else:
            from .adam import GraphAdam
            self.optimizer = GraphAdam(groups, lr=0.0, eps=1e-15)
        from .schedule import get_expon_lr_func
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=pos_lr, lr_final=getattr(training_args, "position_lr_final", 0.0000016) * self.spatial_lr_scale,
            lr_delay_mult=getattr(training_args, "position_lr_delay_mult", 0.01),
            max_steps=getattr(training_args, "position_lr_max_steps", 30_000))

    def update_learning_rate(self, iteration: int):
        """scene/gaussian_model.py:171-177: the schedule drives the group named "xyz" (only).  With a GraphAdam the new
        rate is also uploaded to the device, where a captured step reads it."""
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                group["lr"] = lr
                if hasattr(self.optimizer, "push_lr"):
                    self.optimizer.push_lr()
                return lr

    def reset_opacity(self):
        """scene/gaussian_model.py:212-215 (+ replace_tensor_to_optimizer :261-274): opacities are clamped to <= 0.01 and
        the Adam moments of the opacity group start again from zero.  Done IN PLACE -- the reference swaps in a new
        nn.Parameter, which would invalidate the pointers of a captured step."""
        with torch.no_grad():
            o = torch.clamp(self.get_opacity, max=0.01)
            self._opacity.copy_(torch.log(o / (1.0 - o)))
        st = getattr(self.optimizer, "state", {}).get(self._opacity) if self.optimizer is not None else None
        if st:
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k].zero_()

    @property
    def num_gs(self) -> int:
        return self._xyz.shape[0]

    def oneupSHdegree(self):
        """scene/gaussian_model.py:121-123."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def add_densification_stats(self, viewspace_point_tensor, update_filter, ok=None):
        """scene/gaussian_model.py:410-412, written with masks instead of boolean indexing (no host sync, so the
        step can be captured into a graph).  ok: optional device scalar (0/1) that voids the update."""
        f = update_filter.unsqueeze(1).to(self.denom.dtype)
        if ok is not None:
            f = f * ok
        self.xyz_gradient_accum += torch.norm(viewspace_point_tensor.grad[:, :2], dim=-1, keepdim=True) * f
        self.denom += f

    # ---- visibility -------------------------------------------------------------
    def get_anchor_points(self) -> torch.Tensor:
        """The point of each Gaussian ON its bound face: the face centre, or the barycentric point when `gs_bc`
        is set (AvatarGaussianModel.get_barycentric_3d, scene/avatar_gaussian_model.py:153-159)."""
        P = self._xyz.shape[0]
        dev = self.mesh.v.device
        with torch.no_grad():
            return mesh_bind(self.mesh.v, self.mesh.f, self.binding, torch.zeros(P, 3, device=dev), self._scaling,
                             self._rotation, self.gs_bc)[0]

    def get_visible_mask(self, camera: torch.Tensor, return_first_hit: bool = False):
        """camera [3] (GPU) -> bool [P]: the first mesh triangle hit by the ray camera -> anchor is the Gaussian's own
        face.  Same semantics as scene/avatar_gaussian_model.py:227-263, but on the device (no open3d, no host
        round trip)."""
        return visible_mask(self.mesh.v.detach(), self.mesh.f, self.binding, self.get_anchor_points(), camera,
                            return_first_hit)

    # ---- persistence / initialisation ---------------------------------------
    def find_valid_gaussians(self):
        """scene/mesh_gaussian_model.py:284-289: Gaussians bound to a face listed in `mesh.valid_faces` (all when the
        list is missing / empty) -- as an index tensor instead of the reference's Python list."""
        valid = getattr(self.mesh, "valid_faces", None)
        if valid is None or len(valid) == 0:
            return torch.arange(self.binding.shape[0], device=self.binding.device)
        vf = torch.as_tensor(list(valid) if not torch.is_tensor(valid) else valid, device=self.binding.device).long()
        return torch.nonzero(torch.isin(self.binding, vf)).reshape(-1)

    def save_ply(self, path: str, save_local: bool = False):
        """scene/mesh_gaussian_model.py:251-283.  save_local=True: mesh-frame parameters (`local_point_cloud.ply`) and
        `binding.pkl` beside it; else the world-frame values the getters produce, scaling stored as its log
        (`point_cloud.ply`).  Property order of scene/gaussian_model.py:179-209; only the valid Gaussians are written."""
        from .mesh_io import save_binding
        from .ply_io import save_gaussians
        m = self.find_valid_gaussians()
        with torch.no_grad():
            if save_local:
                xyz, scale, rot = self._xyz[m], self._scaling[m], self._rotation[m]
            else:
                xyz, scale, rot = self.get_xyz[m], torch.log(self.get_scaling)[m], self.get_rotation[m]
            save_gaussians(path, xyz, self._features_dc[m], self._features_rest[m], self._opacity[m], scale, rot)
        if save_local:
            save_binding(os.path.join(os.path.dirname(path), "binding.pkl"), self.binding[m])

    def load_ply(self, path: str):
        """scene/mesh_gaussian_model.py:290-342: parameters become plain tensors (fixed while the mesh is optimised),
        the binding comes from `binding.pkl` beside the PLY, and mesh.v becomes the trainable leaf."""
        from .mesh_io import load_binding
        from .ply_io import load_gaussians
        dev = self.mesh.v.device if self.mesh is not None else "cuda"
        for k, v in load_gaussians(path, self.max_sh_degree, device=dev).items():
            setattr(self, k, v)
        self.active_sh_degree = self.max_sh_degree
        bpath = os.path.join(os.path.dirname(path), "binding.pkl")
        if os.path.exists(bpath):
            self.binding = load_binding(bpath, device=dev)
        elif self.binding is None or self.binding.shape[0] != self._xyz.shape[0]:
            raise FileNotFoundError(f"{bpath}: a mesh-bound point cloud needs its binding.pkl")
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=dev)
        if self.mesh is not None:
            self.mesh.v = nn.Parameter(self.mesh.v.detach().clone().requires_grad_(True))
        self._bound = None

    def save_mesh(self, path: str):
        """scene/mesh_gaussian_model.py:438-441: the template's OBJ (uvs / faces / texture faces) with the current
        vertices."""
        from .mesh_io import write_obj
        out = dict(getattr(self, "template", None) or {"faces": self.mesh.f.detach().cpu().numpy()})
        out["vertices"] = self.mesh.v.detach().cpu().numpy()
        write_obj(out, path)

    def load_mesh(self, path: str, device=None):
        """A registered frame's mesh (scene/scene.py:155-156): vertices replace mesh.v (faces too when the model has
        none yet); the template dictionary is kept for save_mesh."""
        from .mesh_io import read_obj
        d = read_obj(path)
        dev = device or (self.mesh.v.device if self.mesh is not None else self._xyz.device if self._xyz.numel() else "cuda")
        if self.mesh is None:
            self.mesh = SimpleNamespace(v=None, f=torch.from_numpy(d["faces"]).long().to(dev).contiguous())
        self.mesh.v = nn.Parameter(torch.from_numpy(d["vertices"]).to(dev).contiguous())
        self.template = {k: v for k, v in d.items() if getattr(v, "size", 0)}
        self._bound = None

    def init_scaling_from_neighbours(self, points: torch.Tensor) -> torch.Tensor:
        """scales = log(sqrt(clamp_min(distCUDA2(points), 1e-7))) repeated on 3 axes
        (scene/gaussian_model.py:135-136) through the HIP 3-NN kernel."""
        from simple_knn._C import distCUDA2
        dist2 = torch.clamp_min(distCUDA2(points.float()), 0.0000001)
        return torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)

    def parameters(self):
        return [self.mesh.v, self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling,
                self._rotation]
