"""xSlot attention module on the fused HIP kernels -- mirrors sloter/utils/slot_attention.py:9-96 of the reference
(same constructor, parameter names and return values).  As in the reference: `to_q` exists but is unused
(:52-53), there is no softmax (:55-57), `eps` is unused, logits come from the last iteration's updates (:96)."""
import math

import os

import numpy as np
import torch
from torch import nn

from ... import kernels as K
from ...nn_hip import LinearParams as _LinearParams


class _GRUParams(nn.Module):
    """Parameter holder with nn.GRU(d, d)'s names / init (gate order r, z, n)."""

    def __init__(self, dim):
        super().__init__()
        bound = 1.0 / math.sqrt(dim)
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * dim, dim).uniform_(-bound, bound))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * dim, dim).uniform_(-bound, bound))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * dim).uniform_(-bound, bound))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * dim).uniform_(-bound, bound))
        self._g = {}

    def _bind_grad(self, leaf, seg):
        self._g[leaf] = seg.view(getattr(self, leaf).shape)

    def flatten_parameters(self):
        pass


class _ReLUMarker(nn.Module):
    pass


class SlotAttention(nn.Module):
    def __init__(self, num_classes, slots_per_class, dim, iters=3, eps=1e-8, vis=False, vis_id=0, loss_status=1,
                 power=1, to_k_layer=1):
        super().__init__()
        self.num_classes = num_classes
        self.slots_per_class = slots_per_class
        self.num_slots = num_classes * slots_per_class
        self.iters = iters
        # parameter gradients next to the critical chain (bwd); SCOUTER_HEAD_SIDE=0: on the compute stream (A/B)
        self.use_side_stream = K.SIDE_STREAM_DEFAULT and os.environ.get("SCOUTER_HEAD_SIDE", "1") != "0"
        self.eps = eps
        self.scale = dim ** -0.5
        self.loss_status = loss_status
        mu = torch.randn(1, 1, dim).expand(1, self.num_slots, -1)
        sigma = torch.randn(1, 1, dim).abs().expand(1, self.num_slots, -1)   # torch>=2 rejects a signed std (:25)
        self.initial_slots = nn.Parameter(torch.normal(mu, sigma))
        self.to_q = nn.Sequential(_LinearParams(dim, dim))
        mods = [_LinearParams(dim, dim)]
        for _ in range(1, to_k_layer):
            mods += [_ReLUMarker(), _LinearParams(dim, dim)]
        self.to_k = nn.Sequential(*mods)
        self.gru = _GRUParams(dim)
        self.vis = vis
        self.vis_id = vis_id
        self.power = power
        self.dim = dim
        self._g = {}
        self.last_attn = None        # [B, S, N] fp32 attention of the last iteration (the "attention map" artifact)

    def _bind_grad(self, leaf, seg):
        self._g[leaf] = seg.view(self.initial_slots.shape)

    def _to_k_layers(self):
        return [m for m in self.to_k if isinstance(m, _LinearParams)]

    # ---- explicit forward / backward used by SlotModel's fused autograd node
    def fwd(self, X, PE):
        """X: [B, N, d] tokens (post conv1x1+ReLU), PE: [N, d].  Returns the kernel outputs dict (logits, attn,
        area_part + the tensors saved for the backward)."""
        lay = self._to_k_layers()
        g = self.gru
        out = K.xslot_fwd(X, PE, [m.weight for m in lay], [m.bias for m in lay], self.initial_slots[0],
                          g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, self.slots_per_class,
                          self.iters, self.loss_status)
        self.last_attn = out["attn"]
        return out

    def bwd(self, X, PE, saved, dlogits, g_area_sum):
        """Returns dX [B, N, d]; writes the gradients of initial_slots / to_k / gru into the bound arena slices."""
        lay = self._to_k_layers()
        g = self.gru
        T, B, S, d = self.iters, X.shape[0], self.num_slots, self.dim
        r = K.xslot_bwd(X, PE, [m.weight for m in lay], self.initial_slots[0], g.weight_ih_l0, g.weight_hh_l0,
                        g.bias_ih_l0, g.bias_hh_l0, saved, dlogits, g_area_sum, self.slots_per_class, T,
                        self.loss_status)
        # Everything below is parameter gradients -- two dozen 3-10 us launches (GEMMs over the rows the kernel emitted, column
        # sums) that nothing on the way back to the backbone waits for: they go to the weight-gradient side stream like the
        # convolutions' (nn_hip.Conv2d.bwd), and the step's critical chain continues with relu_bwd / conv1x1's input gradient
        # right behind xslot_bwd (round 6: they sat in front of it on the compute stream with the GPU all but idle).
        # SlotModel._fused_backward joins the side stream before anything reads the gradient arena.
        with K.side_stream(X.device, r["ds0"], r["dgi"], r["dgh"], r["U"], r["dZ"], saved["H"], saved["states"],
                           enabled=self.use_side_stream):
            self._param_grads(r, saved, X)
        return r["dX"]

    def _param_grads(self, r, saved, X):
        lay = self._to_k_layers()
        g = self.gru
        T, B, S, d = self.iters, X.shape[0], self.num_slots, self.dim
        if "initial_slots" in self._g:
            K.colsum(r["ds0"].view(B, S * d), self._g["initial_slots"].view(-1))
        if T > 1:
            M = (T - 1) * B * S
            dgi, dgh, U = r["dgi"].view(M, 3 * d), r["dgh"].view(M, 3 * d), r["U"].view(M, d)
            if "weight_ih_l0" in g._g:
                K.matmul_tn(dgi, U, g._g["weight_ih_l0"])
            if "weight_hh_l0" in g._g:
                # dW_hh = sum_t dgh_t^T h_t with the hidden state entering GRU step t: h_0 = the initial slots -- the SAME
                # [S, d] rows for every image, so its term is (sum over the batch of dgh_0)^T s_0: a column sum over B and an
                # S-row product instead of B copies of s_0 (round 4 materialised `sprev` with an expand().contiguous() and one
                # copy per iteration: 60 MB per step at BASELINE configs[4]) -- then the saved states, read where they lie
                gW = g._g["weight_hh_l0"]
                dgh0 = torch.empty((S, 3 * d), dtype=torch.float32, device=X.device)
                K.colsum(r["dgh"][0].view(B, S * 3 * d), dgh0.view(-1))
                if T > 2:
                    K.matmul_tn(dgh[B * S:], saved["states"][:T - 2].view((T - 2) * B * S, d), gW)
                    first = torch.empty_like(gW)
                    K.matmul_tn(dgh0, self.initial_slots[0], first)
                    K.axpby(first, gW, 1.0, 1.0, out=gW)
                else:
                    K.matmul_tn(dgh0, self.initial_slots[0], gW)
            if "bias_ih_l0" in g._g:
                K.colsum(dgi, g._g["bias_ih_l0"])
            if "bias_hh_l0" in g._g:
                K.colsum(dgh, g._g["bias_hh_l0"])
        else:
            for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                if k in g._g:
                    g._g[k].zero_()
        N = X.shape[1]
        for l, m in enumerate(lay):
            dz, hin = r["dZ"][l].view(B * N, d), saved["H"][l].view(B * N, d)
            if "weight" in m._g:
                K.matmul_tn(dz, hin, m._g["weight"])
            if "bias" in m._g:
                K.colsum(dz, m._g["bias"])

    def vis_maps(self, attn=None):
        """The uint8 per-class maps of `--vis true` (slot_attention.py:68-83) for image `vis_id`."""
        a = (self.last_attn if attn is None else attn).detach().float().cpu()
        if self.slots_per_class > 1:
            a = a.reshape(a.shape[0], self.num_classes, self.slots_per_class, a.shape[-1]).sum(2)
        a = a[self.vis_id]
        side = int(a.size(1) ** 0.5)
        span = a.max() - a.min()
        if float(span) == 0.0:
            # a constant map (one token, or every attention equal): the reference divides 0 / 0 and casts the NaNs to uint8,
            # which is platform-defined in NumPy (x86 gives 0, and warns); defined here -- and in the oracle -- as 0
            return np.zeros((a.shape[0], side, side), dtype=np.uint8)
        a = ((a - a.min()) / span * 255.).reshape(a.shape[0], side, side)
        return a.numpy().astype(np.uint8)

    def save_vis(self, folder="sloter/vis"):
        import os
        from PIL import Image
        os.makedirs(folder, exist_ok=True)
        for i, image in enumerate(self.vis_maps()):
            Image.fromarray(image, mode="L").save(os.path.join(folder, "slot_%d.png" % i))

    def forward(self, inputs, inputs_x):
        """Reference signature (slot_attention.py:44): inputs = x + pe, inputs_x = x, both [B, N, d]; returns
        (logits [B, C], area ** power).  Differentiable on its own (training normally goes through SlotModel's fused
        node): one autograd node around the fused kernels.  `inputs - inputs_x` must be the batch-invariant positional
        encoding, as in slot_model.py:110-115; the gradient of both arguments is returned on `inputs_x` (their sum is
        what reaches the common ancestor x)."""
        if torch.is_grad_enabled() and (inputs_x.requires_grad or inputs.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            logits, term = _XSlotFn.apply(self, inputs, inputs_x, self.initial_slots)
        else:
            with torch.no_grad():
                X = inputs_x.float().contiguous()
                out = self.fwd(X, (inputs.float() - X)[0].contiguous())
                B, S, N = out["attn"].shape
                logits, term = out["logits"], torch.pow(out["area_part"].sum() / (B * S * N), self.power)
        if self.vis:
            self.save_vis()
        return logits, term

    def _named_leaves(self):
        """(owner module, leaf name, parameter) of every parameter the backward produces a gradient for"""
        yield self, "initial_slots", self.initial_slots
        for leaf in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            yield self.gru, leaf, getattr(self.gru, leaf)
        for m in self._to_k_layers():
            yield m, "weight", m.weight
            yield m, "bias", m.bias


class _XSlotFn(torch.autograd.Function):
    """Stand-alone autograd node of SlotAttention.forward.  `anchor` (initial_slots) only ties the node to the
    module's parameters; their gradients are written by SlotAttention.bwd into temporary buffers (or the bound arena
    slices) and accumulated into `.grad` here."""

    @staticmethod
    def forward(ctx, mod, inputs, inputs_x, anchor):
        X = inputs_x.detach().float().contiguous()
        PE = (inputs.detach().float() - X)[0].contiguous()
        out = mod.fwd(X, PE)
        B, S, N = out["attn"].shape
        area = out["area_part"].sum() / (B * S * N)
        ctx.mod, ctx.saved, ctx.dims = mod, (X, PE, out, area), (B, S, N)
        return out["logits"], torch.pow(area, mod.power)

    @staticmethod
    def backward(ctx, g_logits, g_term):
        mod = ctx.mod
        X, PE, out, area = ctx.saved
        B, S, N = ctx.dims
        dlogits = (g_logits if g_logits is not None else torch.zeros_like(out["logits"])).float().contiguous()
        g_term = g_term if g_term is not None else torch.zeros((), device=X.device)
        # d(area**p)/d(sum A) = p * area**(p-1) / (B S N)
        g_area_sum = (g_term.float() * mod.power * torch.pow(area, mod.power - 1) / (B * S * N)).reshape(1).contiguous()
        temp = []
        for owner, leaf, p in mod._named_leaves():
            if p.requires_grad and leaf not in owner._g:
                buf = torch.zeros(p.numel(), dtype=torch.float32, device=X.device)
                owner._bind_grad(leaf, buf)
                temp.append((owner, leaf, p))
        try:
            dX = mod.bwd(X, PE, out, dlogits, g_area_sum)
            K.join_side_stream(X.device)
            for owner, leaf, p in temp:
                g = owner._g[leaf].reshape(p.shape).clone()
                p.grad = g if p.grad is None else p.grad + g
        finally:
            for owner, leaf, _ in temp:
                del owner._g[leaf]
        return None, None, dX, None
