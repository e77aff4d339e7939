"""Fused multi-tensor AdamW on the HIP path (one kernel launch per step) with torch.optim.AdamW's interface and
numerics (reference train.py:146: `torch.optim.AdamW(params, lr=args.lr)` -- betas (0.9, 0.999), eps 1e-8,
weight_decay 1e-2; `args.weight_decay` is ignored by the reference and therefore here too).

Gradients are read from the model's flat GradArena (param.grad are views of it); exp_avg / exp_avg_sq live in two
flat arenas with the same offsets, so `step()` is a single launch over a chunk table.

Deviation from torch semantics, by design: every backward of the HIP path OVERWRITES the gradient arena (it does not
accumulate into .grad), so `zero_grad()` has nothing to clear and two backwards before one `step()` leave only the
second gradient -- gradient accumulation is not part of the reference's loop (engine.py:29-35) and is not supported."""
import ctypes
import struct

import torch

from . import _native
from . import kernels as K


class FusedAdamW(torch.optim.Optimizer):
    CHUNK = 16384       # elements per workgroup: ~1000 workgroups for resnest26d (16-byte accesses, 16 per thread)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._plan = None
        # capturable (like torch.optim's flag): learning rate and step count are read by the kernel from a small device
        # buffer, so step() contains no host-computed scalar and can be recorded into a hipGraph (scouter_amd/graph.py).
        # The host mirror state["step"] is still advanced, so state_dict() / resume are unchanged.
        self.capturable = bool(capturable)
        self._dyn = {}                       # group index -> device tensor [lr, step]

    # ---- plan: one chunk table per param group over a private flat layout (offsets 16B-aligned)
    def _build_plan(self):
        plans = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                plans.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamW runs on a HIP device only (no CPU fallback)")
            base = min(p.grad.data_ptr() for p in ps)
            end = max(p.grad.data_ptr() + p.grad.numel() * 4 for p in ps)
            span = (end - base) // 4
            rows = []
            for p in ps:
                g = p.grad
                if g.dtype != torch.float32 or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW: fp32 parameters / gradients only")
                # parameter and gradient must share the same dense physical layout (true for the arena views)
                if g.stride() != p.stride():
                    raise RuntimeError("FusedAdamW: gradient layout differs from the parameter's")
                off = (g.data_ptr() - base) // 4
                n = p.numel()
                for c0 in range(0, n, self.CHUNK):
                    rows.append((p.data_ptr() + 4 * c0, off + c0, min(self.CHUNK, n - c0)))
            blob = b"".join(struct.pack("<QqiI", a, o, n, 0) for a, o, n in rows)
            assert len(blob) == len(rows) * _native.lib().scouter_adamw_chunk_bytes()
            table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            state = self.state.setdefault("_flat_%d" % len(plans), {})
            if "exp_avg" in state and state["exp_avg"].numel() == span and state["exp_avg"].device != dev:
                # restored from a checkpoint loaded with map_location="cpu": keep the moments, move them
                state["exp_avg"] = state["exp_avg"].to(dev, torch.float32).contiguous()
                state["exp_avg_sq"] = state["exp_avg_sq"].to(dev, torch.float32).contiguous()
            if "exp_avg" not in state or state["exp_avg"].numel() != span:
                if "exp_avg" in state:
                    raise RuntimeError("FusedAdamW: restored moment arena has %d elements, the model's gradient arena "
                                       "spans %d -- the checkpoint belongs to another parameter set"
                                       % (state["exp_avg"].numel(), span))
                state["exp_avg"] = torch.zeros(span, dtype=torch.float32, device=dev)
                state["exp_avg_sq"] = torch.zeros(span, dtype=torch.float32, device=dev)
            state.setdefault("step", 0)
            plans.append(dict(table=table, n=len(rows), base=base, span=span, ids=[id(p) for p in ps],
                              gptrs=[p.grad.data_ptr() for p in ps], pptrs=[p.data_ptr() for p in ps], state=state))
        self._plan = plans

    def load_state_dict(self, state_dict):
        """torch's Optimizer.load_state_dict casts per-PARAMETER state to the parameter's device; the flat moment
        arenas live under string keys (`_flat_<group>`) and would stay wherever `torch.load(map_location=...)` put
        them.  Move them next to the parameters and drop the cached launch plan (it references the old state)."""
        super().load_state_dict(state_dict)
        self._plan = None
        self._dyn = {}
        self.__dict__.pop("_lr_pushed", None)
        for gi, group in enumerate(self.param_groups):
            st = self.state.get("_flat_%d" % gi)
            if not st or not group["params"]:
                continue
            dev = group["params"][0].device
            for key in ("exp_avg", "exp_avg_sq"):
                if key in st and torch.is_tensor(st[key]):
                    st[key] = st[key].to(dev, torch.float32).contiguous()
            st["step"] = int(st.get("step", 0))

    def _plan_valid(self):
        if self._plan is None:
            return False
        for gi, (group, plan) in enumerate(zip(self.param_groups, self._plan)):
            ps = [p for p in group["params"] if p.grad is not None]
            if plan is None:
                if ps:
                    return False
                continue
            if [id(p) for p in ps] != plan["ids"] or [p.grad.data_ptr() for p in ps] != plan["gptrs"] \
                    or [p.data_ptr() for p in ps] != plan["pptrs"]:
                return False
            if plan["state"] is not self.state.get("_flat_%d" % gi):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._plan_valid():
            self._build_plan()
        L = _native.lib()
        for gi, (group, plan) in enumerate(zip(self.param_groups, self._plan)):
            if plan is None:
                continue
            st = plan["state"]
            st["step"] += 1
            b1, b2 = group["betas"]
            grads = ctypes.c_void_p(plan["base"])
            if self.capturable:
                dyn = self._dyn_buffer(gi, group, st, advance=True)
                _native.check(L.scouter_adamw_step_dev_f32(plan["table"].data_ptr(), plan["n"], grads,
                                                           st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                           dyn.data_ptr(), float(b1), float(b2), float(group["eps"]),
                                                           float(group["weight_decay"]), K._stream()), "adamw_step_dev")
                continue
            _native.check(L.scouter_adamw_step_f32(plan["table"].data_ptr(), plan["n"], grads,
                                                   st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                   float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                   float(group["weight_decay"]), int(st["step"]),
                                                   K._stream()), "adamw_step")
        return loss

    def _dyn_buffer(self, gi, group, st, advance):
        """Device scalars [lr, step] of a param group.  Created (outside any capture) with step = the host count BEFORE
        this call's increment; `advance` adds 1 on the device (a tiny captured kernel), so replays keep counting."""
        dyn = self._dyn.get(gi)
        if dyn is None:
            dev = st["exp_avg"].device
            dyn = self._dyn[gi] = torch.tensor([float(group["lr"]), float(st["step"] - 1)], dtype=torch.float32, device=dev)
        if advance:
            dyn[1:2].add_(1.0)
        return dyn

    def sync_hyperparameters(self, replays=0):
        """Before a graph replay: push a changed learning rate (lr schedulers act on param_groups on the host) into the
        device scalars; `replays` = steps the captured graph is about to run, added to the host-side step mirror."""
        if not self.capturable:
            return
        for gi, group in enumerate(self.param_groups):
            dyn = self._dyn.get(gi)
            if dyn is None:
                continue
            if getattr(self, "_lr_pushed", {}).get(gi) != float(group["lr"]):
                dyn[0:1].fill_(float(group["lr"]))
                self.__dict__.setdefault("_lr_pushed", {})[gi] = float(group["lr"])
            st = self.state.get("_flat_%d" % gi)
            if st is not None and replays:
                st["step"] += int(replays)

    def zero_grad(self, set_to_none=False):
        """Gradients live in the model's flat arena and are overwritten by every backward; keeping the views in
        place (set_to_none=False by default here) avoids re-planning each step."""
        if set_to_none:
            super().zero_grad(set_to_none=True)
