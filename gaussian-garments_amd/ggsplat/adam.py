"""GraphAdam -- the optimiser of the inner loops on the graph-capturable kernels of libggsplat.so.

Stands where the reference builds ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` (scene/mesh_gaussian_model.py:375,
gaussian_model.py:165, avatar_net.py:50): same `param_groups` shape ({"params", "lr", "name"}), same update
(single-tensor Adam without amsgrad / weight decay), `step()` / `zero_grad()`.  Differences, all so that a whole
optimisation step can be replayed as one hipGraph:
  * step count, bias corrections and the learning rates live in device memory (`push_lr()` uploads the host-side
    `param_groups[i]["lr"]` after a schedule update -- a 4-byte-per-group async copy, no re-capture);
  * moments are allocated up front (nothing is allocated or zero-filled inside a captured step);
  * `step(guard=...)`: a device int64/uint64 word that voids the whole step when non-zero -- the binning-overflow
    flag of the forward of the same step (ggsplat.rasterizer.last_header()[1:2]).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch

from ._lib import check, lib, ptr, stream_ptr


class _LrStaging:
    """Pinned staging rows of GraphAdam.push_lr, used round robin, each with the event of its last upload: with iterations in
    flight (PipelinedRegistrationStep has no per-iteration host sync) the next push must not rewrite a row whose copy has not
    run yet.  A deep copy of the optimiser (tests clone whole models) gets fresh rows: events are not copyable."""
    ROWS = 4

    def __init__(self, n_groups: int):
        self.n = n_groups
        self.rows = torch.zeros(self.ROWS, n_groups, dtype=torch.float32).pin_memory()
        self.events = [None] * self.ROWS
        self.slot = 0

    def __deepcopy__(self, memo):
        return _LrStaging(self.n)

    def next_row(self) -> torch.Tensor:
        k = self.slot
        self.slot = (k + 1) % self.ROWS
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy that last read this row (four pushes ago) has run
        self._k = k
        return self.rows[k]

    def uploaded(self, dev) -> None:
        ev = self.events[self._k] = self.events[self._k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))


class GraphAdam:
    def __init__(self, params: Iterable, lr: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-15):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.param_groups: List[Dict] = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            self.param_groups.append(g)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        first = self.param_groups[0]["params"][0]
        if first.device.type != "cuda":
            raise RuntimeError("ggsplat GraphAdam runs on the GPU only (no CPU path in the product)")
        dev = self.device = first.device
        L = lib()
        self._state_bytes = int(L.ggs_adam_state_bytes())
        self._lr_dev = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        self._lr_host = _LrStaging(len(self.param_groups))
        self.state: Dict[torch.Tensor, Dict[str, torch.Tensor]] = {}
        for g in self.param_groups:
            for p in g["params"]:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("GraphAdam: parameters must be contiguous float32 tensors")
                self.state[p] = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p),
                                 # step count + bias corrections of THIS parameter (torch counts steps per parameter)
                                 "state": torch.zeros(self._state_bytes, dtype=torch.uint8, device=dev)}
        self.push_lr()

    # ---- learning rates ---------------------------------------------------------------------------------
    def push_lr(self) -> None:
        """Upload param_groups[i]["lr"] to the device (call after changing a learning rate, outside a capture).  Asynchronous:
        the values take effect for the work queued AFTER this call on the current stream."""
        row = self._lr_host.next_row()
        for i, g in enumerate(self.param_groups):
            row[i] = float(g["lr"])
        self._lr_dev.copy_(row, non_blocking=True)
        self._lr_host.uploaded(self.device)

    @property
    def step_count(self) -> int:
        """Steps taken by the most-stepped parameter."""
        return max((int(st["state"][:8].view(torch.int64).item()) for st in self.state.values()), default=0)

    # ---- checkpoints (the reference's capture() / restore() store optimizer.state_dict(), scene/gaussian_model.py:63-90) --
    def state_dict(self) -> Dict:
        """torch.optim.Optimizer.state_dict() layout: parameters numbered in group order, per-parameter state {step, exp_avg,
        exp_avg_sq}, param_groups with "params" = those numbers."""
        idx, state, groups = 0, {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                st = self.state[p]
                state[idx] = {"step": st["state"][:8].view(torch.int64)[0].to(torch.float32).clone(),
                              "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
                ids.append(idx)
                idx += 1
            # the keys torch.optim.Adam keeps per group, with the values this optimiser implements: a torch.optim.Adam built
            # over the same parameters can load_state_dict() this dictionary and continue (its loader REPLACES the groups)
            groups.append({"weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                           "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                           **{k: v for k, v in g.items() if k != "params"}, "betas": self.betas, "eps": self.eps, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: Dict) -> None:
        """Accepts its own state_dict() and torch.optim.Adam's (same layout; amsgrad / weight decay / maximize are refused:
        this optimiser does not implement them).  The bias corrections are rebuilt from the step count with Python pow, while
        the kernel keeps running products: a restored optimiser continues within one ulp of the original per step."""
        for gs in sd["param_groups"]:
            if gs.get("amsgrad") or gs.get("maximize") or gs.get("weight_decay"):
                raise ValueError("GraphAdam.load_state_dict: amsgrad / maximize / weight_decay are not implemented")
        flat = [p for g in self.param_groups for p in g["params"]]
        for i, st in sd["state"].items():
            p = flat[int(i)]
            mine = self.state[p]
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            # the device-side AdamState {i64 step, f32 1 - b1^t, f32 sqrt(1 - b2^t), f64 b1^t, f64 b2^t} (csrc/ggs_adam.hip)
            n = int(float(st["step"]))
            b1, b2 = self.betas
            import struct
            raw = struct.pack("<qffdd", n, 1.0 - b1 ** n, (1.0 - b2 ** n) ** 0.5, b1 ** n, b2 ** n)
            mine["state"][:32].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = gs["lr"]
        self.push_lr()

    # ---- torch.optim.Optimizer surface ------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def tick_states(self, params=None) -> List[int]:
        """Device addresses of the states step() would advance if every parameter of `params` (default: all) had a gradient --
        for a caller that advances them itself (ggs_registration_aux_tail) and then calls step(tick=False)."""
        ps = [p for g in self.param_groups for p in g["params"]] if params is None else list(params)
        return [self.state[p]["state"].data_ptr() for p in ps if p.numel()]

    @torch.no_grad()
    def step(self, guard: Optional[torch.Tensor] = None, tick: bool = True) -> None:
        """tick=False: the step counts / bias corrections of the tensors that have a gradient were advanced already."""
        L = lib()
        stream = stream_ptr(self.device)
        if guard is not None and (guard.element_size() != 8 or guard.device != self.device):
            raise ValueError("GraphAdam.step: guard must be a 64-bit word on the optimiser's device")
        gp = ptr(guard)
        b1, b2 = self.betas
        # every tensor that has a gradient: one launch per 16 tensors
        items = []
        for i, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                grad = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.float().contiguous()
                st = self.state[p]
                items.append((p.numel(), p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(),
                              st["exp_avg_sq"].data_ptr(), self._lr_dev.data_ptr() + 4 * i, st["state"].data_ptr(), grad))
        for k in range(0, len(items), 16):
            chunk = items[k:k + 16]
            n = len(chunk)
            numel = (C.c_size_t * n)(*[c[0] for c in chunk])
            cols = [(C.c_void_p * n)(*[c[j] for c in chunk]) for j in range(1, 7)]
            # step counts / bias corrections advance inside the update launch (the last workgroup of a tensor stores them)
            fn = L.ggs_adam_tick_step_multi if tick else L.ggs_adam_step_multi
            check(fn(n, numel, cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], b1, b2, self.eps, gp, stream),
                  "ggs_adam_tick_step_multi" if tick else "ggs_adam_step_multi")
