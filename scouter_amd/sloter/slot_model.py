"""SlotModel on the MI355X HIP path -- mirrors sloter/slot_model.py:10-127 of the reference (same constructor
`args`, attributes, state_dict keys and return structure: `log_probs` or `[log_probs, [loss, nll, area_loss]]`).

The whole step (backbone -> conv1x1+ReLU -> sine PE -> fused xSlot -> log_softmax/NLL + lambda*area) is ONE autograd
node: the forward launches the HIP kernels layer by layer and records what each layer's hand-written backward
needs; `loss.backward()` then runs those backward kernels in reverse and writes every parameter gradient into a
flat arena (`param.grad` are views of it).  There is no ATen arithmetic on the path and no CPU fallback."""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from .. import kernels as K
from ..nn_hip import Conv2d, GradArena, StemConv2d
from ..timm.models import create_model
from .utils.position_encode import build_position_encoding
from .utils.slot_attention import SlotAttention

# bf16 activation storage under --precision bf16 (SlotModel.set_activation_storage); SCOUTER_BF16_STORAGE=0: fp32 storage
BF16_STORAGE_DEFAULT = os.environ.get("SCOUTER_BF16_STORAGE", "1") != "0"
# the per-step weight-plane split on its own stream next to the stem (SlotModel._forward_impl); 0: on the compute stream
SPLIT_ASYNC = os.environ.get("SCOUTER_SPLIT_ASYNC", "1") != "0"


class Identical(nn.Module):
    def forward(self, x):
        return x


def load_backbone(args):
    """reference slot_model.py:18-52 (resnet-family branch; the other timm families are out of scope)."""
    # slot_model.py:19-22: a 3-channel model (pretrained weights when args.pre_trained: a LOCAL file, see
    # timm/models/helpers.py -- raises when it is absent rather than freezing random layers), then the MNIST stem swap
    bone = create_model(args.model, pretrained=bool(getattr(args, "pre_trained", False)), num_classes=args.num_classes)
    if args.dataset == "MNIST":
        from ..nn_hip import StemConv2d
        bone.conv1 = StemConv2d(1, 64, 3, 2, 1)                 # slot_model.py:23-24
    if args.use_slot:
        if getattr(args, "use_pre", False):
            checkpoint = torch.load(f"saved_model/{args.dataset}_no_slot_checkpoint.pth", map_location="cpu", weights_only=False)
            new_state_dict = OrderedDict((k[9:], v) for k, v in checkpoint["model"].items())   # strip `backbone.`
            bone.load_state_dict(new_state_dict)
            print("load pre dataset parameter over")
        if not getattr(args, "grad", False):
            if "res" not in args.model:
                raise RuntimeError("only the resnet / resnest backbones are built")
            bone.global_pool = Identical()
            bone.fc = Identical()
    return bone


class _FusedStep(torch.autograd.Function):
    """The single autograd node of the model.  `anchor` is a dummy leaf that ties the outputs into the graph;
    parameter gradients are delivered through the flat arena (param.grad views), not through autograd edges."""

    @staticmethod
    def forward(ctx, anchor, model, x, target):
        ctx.set_materialize_grads(False)
        logp, stats, state = model._forward_impl(x, target, save=True)
        ctx.model, ctx.state = model, state
        return logp, stats[0], stats[1], stats[2]

    @staticmethod
    def backward(ctx, g_logp, g_loss, g_nll, g_term):
        ctx.model._backward_impl(ctx.state, g_logp, g_loss, g_nll, g_term)
        ctx.state = None
        return None, None, None, None


class SlotModel(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.use_slot = args.use_slot
        # "fp32" (parity path) or "bf16": convolution matrix inputs rounded to bf16, fp32 accumulation; the wide bottleneck
        # tensors stored as bf16 (set_activation_storage)
        self.precision = str(getattr(args, "precision", "fp32"))
        if self.precision not in ("fp32", "bf16"):
            raise ValueError("precision must be fp32 or bf16, got %r" % self.precision)
        self.backbone = load_backbone(args)
        self.set_precision(self.precision)
        self.set_planes(int(os.environ.get("SCOUTER_PLANES", "3")))
        self.set_x3(K.X3_DEFAULT)
        self._arena = None
        self._anchor = None
        self._split_device = None
        self._flatten_tracked()              # BatchNorm counters: views of one flat buffer, from construction on
        self._post_backward_hooks = []       # called with the GradArena after every backward (data-parallel reduce)
        self._grad_ready_hooks = []          # called (arena, lo, hi) as soon as arena.flat[lo:hi] is final
        self.last_stats = None               # device tensor [loss, nll, area**power, top1, area] of the last forward
        if not self.use_slot:                # FC baseline (slot_model.py:75-77): backbone + global pool + fc
            if args.pre_trained:
                self.dfs_freeze(self.backbone, args.freeze_layers)
            return
        self.feature_size = 9            # kept for attribute compatibility; the grid is derived from the features
        self.channel = args.channel
        self.slots_per_class = args.slots_per_class
        self.conv1x1 = Conv2d(self.channel, args.hidden_dim, 1, 1, 0, bias=True)
        if args.pre_trained:
            self.dfs_freeze(self.backbone, args.freeze_layers)
        self.slot = SlotAttention(args.num_classes, self.slots_per_class, args.hidden_dim, vis=args.vis,
                                  vis_id=args.vis_id, loss_status=args.loss_status, power=args.power,
                                  to_k_layer=args.to_k_layer)
        self.position_emb = build_position_encoding("sine", hidden_dim=args.hidden_dim)
        self.lambda_value = float(args.lambda_value)

    def set_precision(self, precision):
        """Matrix-input precision of the BACKBONE convolutions (the xSlot head incl. conv1x1 always runs in fp32).
        Stored on each layer object, so two models of different precision never share a switch."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be fp32 or bf16, got %r" % precision)
        self.precision = precision
        for mod in self.backbone.modules():
            if isinstance(mod, Conv2d):
                mod.precision = precision
        self.set_activation_storage("bf16" if precision == "bf16" and BF16_STORAGE_DEFAULT else "fp32")

    def set_activation_storage(self, storage):
        """"bf16" (the default under precision "bf16"; SCOUTER_BF16_STORAGE=0 turns it off): every ResNeSt bottleneck but
        the last stores its 4x-wide tensors -- conv3 / downsample-convolution outputs and the block output -- as bf16 in
        HBM (half the bytes of the passes that dominate BASELINE configs[4]); arithmetic and every gradient stay fp32.
        The oracle emulates it with oracle.torch_oracle.ACTIVATION_STORAGE."""
        if storage not in ("fp32", "bf16"):
            raise ValueError("activation storage must be fp32 or bf16, got %r" % storage)
        if storage == "bf16" and self.precision != "bf16":
            raise ValueError("bf16 activation storage is an option of precision 'bf16'")
        from ..timm.models.resnest import ResNestBottleneck, GRAD_STREAM_BF16_DEFAULT
        blocks = [m for m in self.backbone.modules() if isinstance(m, ResNestBottleneck)]
        for i, blk in enumerate(blocks):
            blk.store_bf16 = storage == "bf16" and i + 1 < len(blocks)
            # the residual-stream gradient follows the activation storage (SCOUTER_BF16_GRAD_STREAM=0: fp32 under bf16 storage)
            blk.grad_stream_bf16 = storage == "bf16" and GRAD_STREAM_BF16_DEFAULT
        self.activation_storage = storage if blocks else "fp32"

    def set_planes(self, nplanes):
        """3 (default): the grouped 3x3 convolutions of the split-attention blocks run on the bf16 matrix cores over
        exact three-way bf16 splits of their fp32 operands (csrc/conv_planes.hip: six products per fp32 product, fp32
        accumulation -- the accuracy of the exact-fp32 MFMA kernel at 1.6x its speed); 0: every convolution on the fp32
        MFMA kernels.  Only layers whose shapes qualify switch (Conv2d.planes_in / planes_dy).  With precision="bf16" the
        same layers take ONE plane -- the RNE-rounded bf16 value the bf16-input kernels would form in flight -- from their
        producers: half the operand bytes, LDS-DMA instead of a conversion pass through registers."""
        if nplanes not in (0, 3):
            raise ValueError("planes must be 0 or 3")
        from ..timm.models.layers.split_attn import SplitAttnConv2d
        for mod in self.backbone.modules():
            if isinstance(mod, SplitAttnConv2d):
                mod.conv.planes = nplanes
        self._plane_convs = [m for m in self.backbone.modules() if isinstance(m, Conv2d) and m.planes]
        self._wsplitter = K.PlaneWeightSplitter()
        self._refresh_x3()

    def set_x3(self, bits):
        """Deep pointwise (1x1) convolutions of the backbone on the register-split bf16x3 GEMM (csrc/conv_x3.hip: fp32
        tensors in and out, fp32-grade products on the bf16 matrix cores): bit 0 forward, bit 1 plain input gradient, bit 2
        input gradient with the fused BatchNorm-backward epilogue, bit 3 weight gradient, bit 4 the forward of the 3x3
        layers with 32 input channels per group (kernels.x3_conv_eligible: the stem's 32 -> 64 convolution), bit 5 the forward of
        the short-K pointwise layers on the persistent bf16x3 kernel (nn_hip.Conv2d.xpw_static, csrc/conv_pw_persist_x3.h), bit 6
        the INPUT GRADIENT of the 3x3 layers with 32 input channels per group on the persistent resident-rows kernel
        (nn_hip.Conv2d.halo_dgrad, csrc/conv_xhalo.hip: both deep-stem convolutions and layer1's radix convolutions, plane
        layers whose gradient planes would need 64), bit 7 (opt-in) the FORWARD of the 3x3 layers with 32 output channels per
        group on the same kernel (the stem's 32 -> 32); 0: the exact-fp32 MFMA kernels.  Which layers qualify is a static function of their channels (kernels.x3_eligible,
        kernels.xpw_fwd_eligible), so the forward does not depend on batch or timing."""
        if bits & ~255:
            raise ValueError("x3 bits must be within 0..255")
        for mod in self.backbone.modules():
            if isinstance(mod, Conv2d) and not isinstance(mod, StemConv2d):
                mod.x3 = int(bits)
        self._refresh_x3()

    def _refresh_x3(self):
        self._x3_convs = [m for m in self.backbone.modules() if isinstance(m, Conv2d) and not isinstance(m, StemConv2d) and
                          not m.planes and ((getattr(m, "x3", 0) & 15 and m.x3_static()) or
                                            (getattr(m, "x3", 0) & 16 and m.x3_conv_static()) or
                                            (getattr(m, "x3", 0) & 192 and any(m.x3_halo_static())))]

    def set_side_stream(self, enabled):
        """Weight gradients on the side stream (default) or serially on the compute stream (per-kernel timing)."""
        for mod in self.modules():
            if isinstance(mod, Conv2d) or hasattr(mod, "_param_grads"):        # (+ the xSlot head's parameter gradients)
                mod.use_side_stream = bool(enabled)

    def dfs_freeze(self, model, freeze_layer_num):
        """reference slot_model.py:79-94"""
        if freeze_layer_num == 0:
            return
        unfreeze_layers = ["layer4", "layer3", "layer2", "layer1"][:4 - freeze_layer_num]
        for name, child in model.named_children():
            if any(u in name for u in unfreeze_layers):
                continue
            for param in child.parameters():
                param.requires_grad = False
            self.dfs_freeze(child, freeze_layer_num)

    # ------------------------------------------------------------------------------------------------------------
    def grad_arena(self):
        if self._arena is None or not self._arena.matches(self):
            self._arena = GradArena(self)
        return self._arena

    def _fc_forward(self, feat, target, save):
        logits, cctx = self.backbone.classifier_fwd(feat, save)
        if target is not None and target.dtype != torch.int64:
            target = target.long()
        logp, stats = K.slot_loss_fwd(logits, target, None, 1.0, 0.0, 1.0)      # slot_model.py:117,124
        self.last_stats = stats
        return logp, stats, ((cctx, logp, stats, target) if save else None)

    def _fc_backward(self, hstate, g_logp, g_loss, g_nll, g_term, need_dfeat=True):
        cctx, logp, stats, target = hstate
        f32 = lambda g: None if g is None else g.float().contiguous()
        dlogits, _ = K.slot_loss_bwd(logp, target, stats, f32(g_loss), f32(g_nll), None, f32(g_logp), 1.0, 0.0, 1.0)
        return self.backbone.classifier_bwd(dlogits, cctx, need_dfeat)

    def _head_forward(self, feat, target, save):
        """feat: NHWC backbone features [B, h, w, channel] -> (log_probs, stats, head state)."""
        if not self.use_slot:
            return self._fc_forward(feat, target, save)
        if feat.shape[-1] != self.channel:
            raise RuntimeError("backbone produced %d channels, args.channel is %d" % (feat.shape[-1], self.channel))
        xmap, cctx = self.conv1x1.fwd(feat, save, relu=True)                  # slot_model.py:108-109
        B, h, w, d = xmap.shape
        PE = self.position_emb.table(h, w, feat.device)                       # slot_model.py:110-111
        so = self.slot.fwd(xmap.view(B, h * w, d), PE)                        # slot_model.py:116
        S, N = self.slot.num_slots, h * w
        if target is not None and target.dtype != torch.int64:
            target = target.long()
        logp, stats = K.slot_loss_fwd(so["logits"], target, so["area_part"], B * S * N, self.lambda_value,
                                      self.slot.power)                        # slot_model.py:117-121
        self.last_stats = stats
        if self.slot.vis:
            self.slot.save_vis()
        return logp, stats, ((cctx, xmap, PE, so, logp, stats, target) if save else None)

    def _head_backward(self, hstate, g_logp, g_loss, g_nll, g_term, need_dfeat=True, post=None):
        """post: the last backbone block's K.BnBwdFuse -- conv1x1's input-gradient epilogue finishes its reductions."""
        if not self.use_slot:
            return self._fc_backward(hstate, g_logp, g_loss, g_nll, g_term, need_dfeat)
        cctx, xmap, PE, so, logp, stats, target = hstate
        B, h, w, d = xmap.shape
        S, N = self.slot.num_slots, h * w
        f32 = lambda g: None if g is None else g.float().contiguous()
        dlogits, g_area = K.slot_loss_bwd(logp, target, stats, f32(g_loss), f32(g_nll), f32(g_term), f32(g_logp),
                                          B * S * N, self.lambda_value, self.slot.power)
        dX = self.slot.bwd(xmap.view(B, N, d), PE, so, dlogits, g_area)
        dxr = K.relu_bwd(dX.view(B, h, w, d), xmap)
        return self.conv1x1.bwd(dxr, cctx, need_dx=need_dfeat, post=post if need_dfeat else None)

    def _forward_impl(self, x, target, save):
        if not x.is_cuda:
            raise RuntimeError("scouter_amd.SlotModel runs on a HIP device (got a %s tensor); there is no CPU fallback "
                               "-- the CPU restatement lives in oracle/ and is test infrastructure only" % x.device)
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        tracked = []
        convs = [c for c in getattr(self, "_plane_convs", ()) if c._nplanes()]
        # (input-gradient planes also where that gradient runs on the resident-rows register-split kernel: Conv2d.halo_dgrad)
        items = [(K.hwio(c.weight), c.groups, True, bool(save and (c.planes_dy() or c.halo_dgrad()))) for c in convs]
        # (the pointwise layers on the register-split GEMM take three WEIGHT planes too -- same launch; fp32 mode only)
        xconvs = [c for c in getattr(self, "_x3_convs", ()) if c.x3_mode()] if (not convs or convs[0]._nplanes() == 3) else []
        # (forward planes only where the forward runs on the register-split GEMM: the persistent bf16x3 forward -- bit 5 -- splits
        #  its weight tile itself)
        xitems = [(c, (bool(c.x3_mode() & 17) and not c.fwd_on_xpw()) or (bool(c.x3_mode() & 64) and c.halo_fwd()),
                   bool(save and ((c.x3_mode() & 6) or ((c.x3_mode() & 64) and c.halo_dgrad())))) for c in xconvs]
        xconvs = [c for c, f, d in xitems if f or d]
        items += [(K.hwio(c.weight), c.groups, f, d) for c, f, d in xitems if f or d]
        if items:      # this step's weight planes of every plane convolution: one launch into persistent buffers ...
            # ... on its own stream NEXT TO the stem's kernels (the stem reads no planes; MFMA-bound next to a byte-moving
            # pass): the backbone joins it after the max-pool (`_post_stem_hooks`).  SCOUTER_SPLIT_ASYNC=0: on the compute stream
            hooks = getattr(self.backbone, "_post_stem_hooks", None)
            use_async = SPLIT_ASYNC and hooks is not None
            if use_async and not hooks:
                hooks.append(lambda: K.join_side_stream(self._split_device, "wsplit") if self._split_device is not None else None)
            self._split_device = x.device if use_async else None
            with K.side_stream(x.device, enabled=use_async, which="wsplit"):
                outs = self._wsplitter.run(items, convs[0]._nplanes() if convs else 3)
            for c, o in zip(convs + xconvs, outs):
                c._wsplit = o
        feat, bctx = self.backbone.features_fwd(x, save, tracked)             # NHWC [B, h, w, channel]
        logp, stats, hstate = self._head_forward(feat, target, save)
        if tracked:
            self._bump_tracked(tracked)                                       # BatchNorm num_batches_tracked
        return logp, stats, ((bctx, hstate) if save else None)

    def loss_seed(self, loss):
        """A cached tensor of ones shaped like the scalar loss: `loss.backward(model.loss_seed(loss))` spares autograd the
        `ones_like` fill it launches for a bare `loss.backward()` (the last ATen kernel of the step; engine.py, bench.py)."""
        seed = getattr(self, "_loss_seed", None)
        if seed is None or seed.device != loss.device or seed.shape != loss.shape or seed.dtype != loss.dtype:
            seed = self._loss_seed = torch.ones_like(loss)
        return seed

    def _flatten_tracked(self):
        """Every BatchNorm's `num_batches_tracked` becomes a view of ONE flat int64 buffer (same buffer names / shapes in the
        state_dict).  Done at construction and after every `_apply` (module.to / .cuda), never during a step: buffer identity is
        stable from the first forward on, so references taken by EMA copies, external buffer lists or a captured hipGraph
        keep pointing at the counters that advance."""
        from ..nn_hip import BatchNorm2d
        bns = [m for m in self.modules() if isinstance(m, BatchNorm2d)]
        self._nbt_runs = {}
        if not bns:
            self._nbt_flat, self._nbt_index = None, {}
            return
        flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in bns]).contiguous()
        for i, m in enumerate(bns):
            m._buffers["num_batches_tracked"] = flat[i]
        self._nbt_flat = flat
        self._nbt_index = {flat[i].data_ptr(): i for i in range(len(bns))}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if hasattr(self, "_nbt_flat"):
            self._flatten_tracked()
        return out

    def _bump_tracked(self, tracked):
        """num_batches_tracked += 1 for the train-mode BatchNorms of this forward (one launch of the library per contiguous run of
        counters in the flat buffer -- ONE when every BatchNorm trains -- instead of an ATen multi-tensor kernel); a BatchNorm
        that ran twice in the forward is counted twice."""
        key = tuple(t.data_ptr() for t in tracked)
        runs = self._nbt_runs.get(key)
        if runs is None:
            if any(p not in self._nbt_index for p in key):
                # a counter was re-assigned behind the model's back (module.num_batches_tracked = ...): flatten again and follow
                # this forward's tensors to their modules
                from ..nn_hip import BatchNorm2d
                by_ptr = {m.num_batches_tracked.data_ptr(): m for m in self.modules() if isinstance(m, BatchNorm2d)}
                mods = [by_ptr[p] for p in key]
                self._flatten_tracked()
                idx = [self._nbt_index[m.num_batches_tracked.data_ptr()] for m in mods]
            else:
                idx = [self._nbt_index[p] for p in key]
            count = {}
            for i in idx:
                count[i] = count.get(i, 0) + 1
            runs, order = [], sorted(count)
            for i in order:
                if runs and runs[-1][1] == i and runs[-1][2] == count[i]:
                    runs[-1][1] = i + 1
                else:
                    runs.append([i, i + 1, count[i]])
            self._nbt_runs[key] = runs
        for lo, hi, c in runs:
            K.iadd_i64(self._nbt_flat[lo:hi], c)

    def _backward_impl(self, state, g_logp, g_loss, g_nll, g_term):
        bctx, hstate = state
        arena = self.grad_arena()
        need = self.backbone._first_trainable_stage() < 5
        last = self.backbone.last_fuse(bctx) if need and self.use_slot else None
        dfeat = self._head_backward(hstate, g_logp, g_loss, g_nll, g_term, need_dfeat=need, post=last)
        # gradient ranges become final from the END of the arena (head, layer4) towards its start (stem)
        done_hi = [arena.numel]

        def stage_done(name):
            lo = arena.first_offset("backbone." + name + ".")
            if lo is not None and lo < done_hi[0]:
                K.join_side_stream(arena.flat.device)       # this stage's weight gradients ran on the side stream
                for hook in self._grad_ready_hooks:
                    hook(arena, lo, done_hi[0])
                done_hi[0] = lo
        if need:
            self.backbone.features_bwd(dfeat, bctx, stage_done if self._grad_ready_hooks else None, own=last)
        K.join_side_stream(arena.flat.device)               # all weight gradients are in the arena from here on
        if self._grad_ready_hooks and done_hi[0] > 0:
            for hook in self._grad_ready_hooks:
                hook(arena, 0, done_hi[0])
        arena.attach()
        for hook in self._post_backward_hooks:
            hook(arena)

    def forward(self, x, target=None):
        train_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if train_graph:
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            self.grad_arena()
            output, loss, nll, attn_loss = _FusedStep.apply(self._anchor, self, x, target)
        else:
            output, stats, _ = self._forward_impl(x, target, save=False)
            loss, nll, attn_loss = stats[0], stats[1], stats[2]
        if target is not None:
            return [output, [loss, nll, attn_loss]] if self.use_slot else [output, [loss]]
        return output
