"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of SCOUTER's xSlot hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file, and
only as the checker / the timed CPU baseline.  Nothing under ``scouter_amd/`` imports it.

What is restated (reference = wbw520/scouter, paths relative to /root/reference):
  * sine positional encoding          sloter/utils/position_encode.py:26-46, :77-81
  * xSlot attention module            sloter/utils/slot_attention.py:44-96 (params :20-38)
  * SlotModel head + loss             sloter/slot_model.py:105-127
  * backbones resnet18/resnest26d/50d timm/models/resnet.py:380-509,134-199,273-306;
                                      timm/models/resnest.py:58-143; timm/models/layers/split_attn.py:14-80
  * optimizer step                    train.py:146 (torch.optim.AdamW defaults)
  * engine step                       engine.py:24-42

The reference is pure Python on top of PyTorch; its arithmetic lives in the un-vendored dependency
``torch==1.6.0`` (requirements.txt:28).  This restatement therefore also computes with torch CPU ops (the
container's torch 2.10 stands in for 1.6), but is written functionally over a flat ``{state_dict key: tensor}``
dict, spells out the GRU cell / split attention / AdamW maths explicitly instead of instantiating nn.GRU /
timm / torch.optim, and works in fp32 or fp64.

PINNING: the reference has no tests and no golden vectors.  The restatement is pinned by
``tests/test_oracle_vs_reference.py`` (runs where /root/reference exists) and by the fixtures under
``tests/golden/`` that ``oracle/gen_golden.py`` produced by importing the reference itself.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------
# architecture tables (timm/models/resnet.py:512-521, timm/models/resnest.py:161-189)
# --------------------------------------------------------------------------------------------------------------
ARCHS = {
    "resnet18": dict(kind="basic", layers=(2, 2, 2, 2), channel=512),
    "resnest26d": dict(kind="resnest", layers=(2, 2, 2, 2), channel=2048),
    "resnest50d": dict(kind="resnest", layers=(3, 4, 6, 3), channel=2048),
}


# --------------------------------------------------------------------------------------------------------------
# positional encoding
# --------------------------------------------------------------------------------------------------------------
def posenc_sine(h, w, d, dtype=torch.float32):
    """PositionEmbeddingSine(num_pos_feats=d/2, normalize=True) for an (h, w) grid -> [d, h, w].

    position_encode.py:26-46: cumsum of an all-true mask gives 1..h / 1..w; normalised by (last + 1e-6) * 2pi;
    divided by 1e4 ** (2*(k//2)/npf); even k -> sin, odd k -> cos; channels 0..npf-1 from y, npf..d-1 from x.
    The reference always computes in fp32 and casts to the feature dtype (:46)."""
    npf = d // 2
    eps, scale, temperature = 1e-6, 2 * math.pi, 10000
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    y_embed = y_embed / (y_embed[-1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).permute(2, 0, 1).to(dtype).contiguous()


# --------------------------------------------------------------------------------------------------------------
# xSlot
# --------------------------------------------------------------------------------------------------------------
def gru_cell(u, h, w_ih, w_hh, b_ih, b_hh):
    """One step of nn.GRU(d, d) (slot_attention.py:38,61-64), gate order r, z, n."""
    gi = u @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    i_r, i_z, i_n = gi.chunk(3, dim=-1)
    h_r, h_z, h_n = gh.chunk(3, dim=-1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1.0 - z) * n + z * h


def num_to_k_layers(P, prefix="slot."):
    L = 0
    while f"{prefix}to_k.{2 * L}.weight" in P:
        L += 1
    return L


def to_k_mlp(P, x_pe, prefix="slot."):
    """slot_attention.py:30-37,47: Linear (ReLU Linear)*"""
    L = num_to_k_layers(P, prefix)
    k = x_pe
    for l in range(L):
        k = F.linear(k, P[f"{prefix}to_k.{2 * l}.weight"], P[f"{prefix}to_k.{2 * l}.bias"])
        if l < L - 1:
            k = torch.relu(k)
    return k


def xslot_forward(P, x_pe, x, num_classes, slots_per_class, loss_status=1, power=1, iters=3, prefix="slot.",
                  aux=None):
    """SlotAttention.forward (slot_attention.py:44-96).  x_pe, x: [B, N, d].  Returns (logits[B,C], area**power).

    As written in the reference: q = slots (to_q unused, :52-53); no softmax; attn = sigmoid(dots / rowsum *
    total) (:55-57); updates use x without PE (:58-59); logits come from the last iteration's updates (:96)."""
    b, n, d = x_pe.shape
    S = num_classes * slots_per_class
    k = to_k_mlp(P, x_pe, prefix)
    slots = P[prefix + "initial_slots"].expand(b, -1, -1)
    scale = d ** -0.5
    w_ih, w_hh = P[prefix + "gru.weight_ih_l0"], P[prefix + "gru.weight_hh_l0"]
    b_ih, b_hh = P[prefix + "gru.bias_ih_l0"], P[prefix + "gru.bias_hh_l0"]
    slot_states = []
    for _ in range(iters):
        slots_prev = slots
        slot_states.append(slots_prev)
        dots = torch.einsum("bid,bjd->bij", slots, k) * scale
        row = dots.sum(2, keepdim=True)                 # r_i
        tot = dots.sum(2).sum(1).view(b, 1, 1)          # tau
        dots = torch.div(dots, row) * tot
        attn = torch.sigmoid(dots)
        updates = torch.einsum("bjd,bij->bid", x, attn) / d
        slots = gru_cell(updates.reshape(-1, d), slots_prev.reshape(-1, d), w_ih, w_hh, b_ih, b_hh)
        slots = slots.reshape(b, -1, d)
    if aux is not None:
        aux["attn"] = attn
        aux["slot_states"] = slot_states
        aux["k"] = k
        aux["updates"] = updates
    if slots_per_class > 1:
        updates = updates.reshape(b, num_classes, slots_per_class, d).sum(2)   # :87-91
    slot_loss = attn.sum() / b / S / n                                         # :93-94 (relu is a no-op)
    return loss_status * updates.sum(2), torch.pow(slot_loss, power)


def vis_maps(attn, num_classes, slots_per_class, vis_id=0):
    """The uint8 attention maps of `--vis true` (slot_attention.py:68-83), without the PNG writing."""
    a = attn
    if slots_per_class > 1:
        a = a.reshape(a.shape[0], num_classes, slots_per_class, a.shape[-1]).sum(2)
    a = a[vis_id]
    side = int(a.size(1) ** 0.5)
    span = a.max() - a.min()
    if float(span) == 0.0:      # constant map: the reference's 0 / 0 -> NaN -> uint8 cast is platform-defined; pinned to 0
        return np.zeros((a.shape[0], side, side), dtype=np.uint8)       # (what the reference's own PNGs hold on x86)
    a = ((a - a.min()) / span * 255.0).reshape(a.shape[0], side, side)
    return a.detach().cpu().numpy().astype(np.uint8)


# Test instrumentation: callable(name, pre_activation) -> activation, applied at every ReLU that follows a BatchNorm /
# closes a residual block / follows conv1x1 (name = the state_dict prefix of that layer, e.g. "backbone.layer2.0.bn1",
# "conv1x1").  None = torch.relu.  tests/test_model_gpu.py uses it to capture the sign pattern, and to evaluate the
# oracle UNDER THE HIP PATH'S sign pattern so that gradient comparisons are not dominated by a ReLU flipping on a ~0
# pre-activation (one flip moves every upstream gradient by ~1/samples in either implementation).
RELU_HOOK = None


def _relu(name, x):
    return torch.relu(x) if RELU_HOOK is None else RELU_HOOK(name, x)


# Same idea for the stem's MaxPool2d(3, 2, 1): callable(x) -> pooled, e.g. maxpool_with_indices under the HIP path's
# window choice (after a ReLU most windows hold several exact zeros, and one flipped sign re-routes the gradient).
MAXPOOL_HOOK = None


def maxpool_with_indices(x, arg):
    """3x3 / stride 2 / pad 1 pooling that takes element `arg` (= ky * 3 + kx, [B, C, Ho, Wo]) of every window."""
    B, C, H, W = x.shape
    cols = F.unfold(F.pad(x, (1, 1, 1, 1), value=float("-inf")), 3, stride=2).view(B, C, 9, -1)
    return cols.gather(2, arg.reshape(B, C, 1, -1).long()).view(B, C, arg.shape[2], arg.shape[3])


def head_forward(P, feat, target, cfg, aux=None):
    """SlotModel.forward after the backbone (slot_model.py:108-127).  feat: [B, Cin, h, w]."""
    x = F.conv2d(feat, P["conv1x1.weight"], P["conv1x1.bias"])
    x = _relu("conv1x1", x)
    b, d, h, w = x.shape
    pe = posenc_sine(h, w, d, x.dtype)
    x_pe = x + pe
    x = x.reshape(b, d, -1).permute(0, 2, 1)
    x_pe = x_pe.reshape(b, d, -1).permute(0, 2, 1)
    logits, attn_loss = xslot_forward(P, x_pe, x, cfg["num_classes"], cfg["slots_per_class"],
                                      cfg.get("loss_status", 1), cfg.get("power", 1), aux=aux)
    output = F.log_softmax(logits, dim=1)
    if aux is not None:
        aux["logits"] = logits
        aux["x"] = x
    if target is None:
        return output
    nll = F.nll_loss(output, target)
    loss = nll + float(cfg.get("lambda_value", 1.0)) * attn_loss
    return output, [loss, nll, attn_loss]


# --------------------------------------------------------------------------------------------------------------
# backbones
# --------------------------------------------------------------------------------------------------------------
# ---- mixed-precision emulation (BASELINE configs[4] "bf16"; the reference itself has no AMP).
# CONV_INPUT_ROUNDING = "bf16" makes every BACKBONE convolution round its matrix operands to bfloat16 (round-to-nearest-
# even) before an exact product -- what the build's precision="bf16" mode computes (fp32 accumulation, fp32 storage):
#   forward : conv(rb(x), rb(w))                       when the layer has >= BF16_MIN_PIXELS output pixels
#   dgrad   : conv_transpose(rb(dy), rb(w))            stride-1 layers with >= BF16_MIN_PIXELS input pixels
#   wgrad   : wgrad(rb(x), rb(dy))                     same-size stride-1 layers, per-group channels multiples of 32
# everything else (tiny attention FCs on the pooled vector, strided gradients, the xSlot head) stays exact.
CONV_INPUT_ROUNDING = None
BF16_MIN_PIXELS = 1024
# ACTIVATION_STORAGE = "bf16" (with CONV_INPUT_ROUNDING = "bf16"): what the build's bf16 activation storage computes
# (scouter_amd SlotModel.set_activation_storage) -- in every ResNeSt bottleneck but the network's last one whose output
# has >= BF16_MIN_PIXELS pixels, the outputs of the radix convolution (conv2.conv), of conv3 and of the downsample
# convolution and the block output are STORED rounded to bf16: the BatchNorm after such a convolution takes its batch
# statistics from the unrounded convolution result and normalises the rounded values; the block output is rounded after the ReLU (straight-through for the
# gradient).  The same blocks store the gradient of their output -- masked by the ReLU, i.e. the gradient of
# `out + residual` -- as bf16: it is rounded once where it is formed and both branches (the bn3 backward and the
# shortcut) read the rounded value.  (The other bf16-stored gradients -- in front of conv1 / conv3 / the downsample
# convolution -- hold exactly what the bf16-operand kernels round to anyway: CONV_INPUT_ROUNDING covers them.)
ACTIVATION_STORAGE = None


def _rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, groups):
        B, _, H, W = x.shape
        y = F.conv2d(_rb(x), _rb(w), None, stride, pad, 1, groups)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, groups)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, groups = ctx.cfg
        B, Cin, H, W = x.shape
        Cout, cg, kh, kw = w.shape
        dgrad_bf16 = stride == 1 and B * H * W >= BF16_MIN_PIXELS
        same = stride == 1 and dy.shape[2] == H and dy.shape[3] == W
        wgrad_bf16 = (same and cg % 32 == 0 and (Cout // groups) % 32 == 0 and B * H * W >= BF16_MIN_PIXELS and
                      ((kh == 1 and kw == 1 and pad == 0) or 64 // W + 1 < H))
        dx = torch.nn.grad.conv2d_input(x.shape, _rb(w) if dgrad_bf16 else w, _rb(dy) if dgrad_bf16 else dy, stride, pad,
                                        1, groups)
        dw = torch.nn.grad.conv2d_weight(_rb(x) if wgrad_bf16 else x, w.shape, _rb(dy) if wgrad_bf16 else dy, stride, pad,
                                         1, groups)
        return dx, dw, None, None, None


def _conv(x, w, bias=None, stride=1, pad=0, dilation=1, groups=1):
    """Backbone convolution: F.conv2d, or its bf16-operand emulation (see CONV_INPUT_ROUNDING)."""
    if CONV_INPUT_ROUNDING is None:
        return F.conv2d(x, w, bias, stride, pad, dilation, groups)
    assert CONV_INPUT_ROUNDING == "bf16" and dilation == 1
    Ho = (x.shape[2] + 2 * pad - w.shape[2]) // stride + 1
    Wo = (x.shape[3] + 2 * pad - w.shape[3]) // stride + 1
    if x.shape[0] * Ho * Wo < BF16_MIN_PIXELS:
        return F.conv2d(x, w, bias, stride, pad, dilation, groups)
    y = _RoundedConv.apply(x, w, stride, pad, groups)
    return y if bias is None else y + bias.view(1, -1, 1, 1)



def _bn(P, name, x, training):
    """nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1, batch statistics in training (resnet.py:383)."""
    y = F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"],
                     P[name + ".bias"], training, 0.1, 1e-5)
    if training and (name + ".num_batches_tracked") in P:
        P[name + ".num_batches_tracked"] += 1
    return y


def _rb_ste(t):
    """round to bf16 in the forward, identity in the backward (a storage rounding)"""
    return t + (_rb(t) - t).detach()


class _GradRound(torch.autograd.Function):
    """identity whose incoming gradient is rounded to bf16 (a gradient tensor stored as bf16)"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


def _bn_stored(P, name, x, training):
    """BatchNorm of a convolution output that is STORED as bf16 (ACTIVATION_STORAGE): statistics (and the running-stat
    update) from the unrounded x, the affine map applied to the rounded values."""
    w, b = P[name + ".weight"], P[name + ".bias"]
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            P[name + ".running_mean"].mul_(0.9).add_(0.1 * mean)
            P[name + ".running_var"].mul_(0.9).add_(0.1 * var * (n / max(n - 1, 1)))
        if (name + ".num_batches_tracked") in P:
            P[name + ".num_batches_tracked"] += 1
    else:
        mean, var = P[name + ".running_mean"], P[name + ".running_var"]
    sc = (w * torch.rsqrt(var + 1e-5)).view(1, -1, 1, 1)
    return (_rb_ste(x) - mean.view(1, -1, 1, 1)) * sc + b.view(1, -1, 1, 1)


def _split_attn(P, name, x, training, stored=False):
    """SplitAttnConv2d.forward, radix 2, cardinality 1 (split_attn.py:54-80, RadixSoftmax :20-28).
    stored: the radix convolution's output is STORED as bf16 (ACTIVATION_STORAGE)."""
    x = _conv(x, P[name + ".conv.weight"], None, 1, 1, 1, 2)
    x = _relu(name + ".bn0", (_bn_stored if stored else _bn)(P, name + ".bn0", x, training))
    B, RC, H, W = x.shape
    x5 = x.reshape(B, 2, RC // 2, H, W)
    gap = x5.sum(dim=1).mean(dim=(2, 3), keepdim=True)
    g = _conv(gap, P[name + ".fc1.weight"], P[name + ".fc1.bias"])
    g = _relu(name + ".bn1", _bn(P, name + ".bn1", g, training))
    a = _conv(g, P[name + ".fc2.weight"], P[name + ".fc2.bias"])
    a = a.view(B, 1, 2, -1).transpose(1, 2)
    a = F.softmax(a, dim=1).reshape(B, 2, RC // 2, 1, 1)
    return (x5 * a).sum(dim=1)


def _resnest_block(P, name, x, stride, training, last=False):
    """ResNestBottleneck.forward with avd=True, avd_first=False (resnest.py:111-143); `is_first` is never set
    by ResNet._make_layer so avd pooling exists only when stride > 1 (:76-80).  last: the network's last block (never
    stored as bf16, see ACTIVATION_STORAGE)."""
    out = _relu(name + ".bn1", _bn(P, name + ".bn1", _conv(x, P[name + ".conv1.weight"]), training))
    stored2 = (ACTIVATION_STORAGE == "bf16" and not last and out.shape[0] * out.shape[2] * out.shape[3] >= BF16_MIN_PIXELS)
    out = _split_attn(P, name + ".conv2", out, training, stored2)
    if stride > 1:
        out = F.avg_pool2d(out, 3, stride, padding=1)
    stored = (ACTIVATION_STORAGE == "bf16" and not last and
              out.shape[0] * out.shape[2] * out.shape[3] >= BF16_MIN_PIXELS)
    bn = _bn_stored if stored else _bn
    out = bn(P, name + ".bn3", _conv(out, P[name + ".conv3.weight"]), training)
    residual = x
    if (name + ".downsample.1.weight") in P:          # downsample_avg (resnet.py:292-306)
        if stride > 1:
            residual = F.avg_pool2d(residual, 2, stride, ceil_mode=True, count_include_pad=False)
        residual = bn(P, name + ".downsample.2", _conv(residual, P[name + ".downsample.1.weight"]), training)
    pre = out + residual
    if stored:
        pre = _GradRound.apply(pre)
    y = _relu(name + ".bn3", pre)
    return _rb_ste(y) if stored else y


def _basic_block(P, name, x, stride, training):
    """BasicBlock.forward (resnet.py:172-199) with downsample_conv (resnet.py:273-289)."""
    out = _relu(name + ".bn1", _bn(P, name + ".bn1", _conv(x, P[name + ".conv1.weight"], None, stride, 1), training))
    out = _bn(P, name + ".bn2", _conv(out, P[name + ".conv2.weight"], None, 1, 1), training)
    residual = x
    if (name + ".downsample.0.weight") in P:
        residual = _bn(P, name + ".downsample.1",
                       _conv(x, P[name + ".downsample.0.weight"], None, stride, 0), training)
    return _relu(name + ".bn2", out + residual)


def backbone_features(P, x, arch, training, prefix="backbone."):
    """ResNet.forward_features (resnet.py:491-501) -> [B, Cin, h, w]."""
    cfg = ARCHS[arch]
    if cfg["kind"] == "resnest":   # deep stem 32-32-64 (resnet.py:389-400)
        x = _conv(x, P[prefix + "conv1.0.weight"], None, 2, 1)
        x = _relu(prefix + "conv1.1", _bn(P, prefix + "conv1.1", x, training))
        x = _conv(x, P[prefix + "conv1.3.weight"], None, 1, 1)
        x = _relu(prefix + "conv1.4", _bn(P, prefix + "conv1.4", x, training))
        x = _conv(x, P[prefix + "conv1.6.weight"], None, 1, 1)
    else:                          # 7x7/2 stem, or the MNIST 3x3/2 1-channel stem (slot_model.py:23-24)
        w = P[prefix + "conv1.weight"]
        x = _conv(x, w, None, 2, w.shape[-1] // 2)
    x = _relu(prefix + "bn1", _bn(P, prefix + "bn1", x, training))
    x = F.max_pool2d(x, 3, 2, 1) if MAXPOOL_HOOK is None else MAXPOOL_HOOK(x)
    block = _resnest_block if cfg["kind"] == "resnest" else _basic_block
    for li, nblocks in enumerate(cfg["layers"]):
        for bi in range(nblocks):
            stride = 2 if (li > 0 and bi == 0) else 1
            if cfg["kind"] == "resnest":
                x = block(P, f"{prefix}layer{li + 1}.{bi}", x, stride, training,
                          last=li == len(cfg["layers"]) - 1 and bi == nblocks - 1)
            else:
                x = block(P, f"{prefix}layer{li + 1}.{bi}", x, stride, training)
    return x


def slot_model_forward(P, x, target, cfg, training=True, aux=None):
    """SlotModel.forward (slot_model.py:105-127) with use_slot=True.  The grid is derived from the feature map
    (the reference hard-codes 9x9, :61-64,108)."""
    feat = backbone_features(P, x, cfg["model"], training)
    if aux is not None:
        aux["feat"] = feat
    return head_forward(P, feat, target, cfg, aux=aux)


def fc_model_forward(P, x, target, arch, training=True):
    """SlotModel.forward with use_slot=False: global average pool + fc (slot_model.py:116-125, resnet.py:503-509)."""
    feat = backbone_features(P, x, arch, training)
    logits = F.linear(feat.mean(dim=(2, 3)), P["backbone.fc.weight"], P["backbone.fc.bias"])
    output = F.log_softmax(logits, dim=1)
    if target is None:
        return output
    return output, [F.nll_loss(output, target)]


# --------------------------------------------------------------------------------------------------------------
# optimizer / engine
# --------------------------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW single-tensor update (train.py:146 uses the defaults; args.weight_decay is ignored)."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def trainable_keys(P):
    return [k for k, v in P.items() if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))
            and "to_q" not in k]


class OracleTrainer:
    """engine.calculation's train branch (engine.py:24-42) over the functional model."""

    def __init__(self, P, cfg, lr=1e-4):
        self.P, self.cfg, self.lr, self.t = P, cfg, lr, 0
        self.keys = trainable_keys(P)
        self.m = {k: torch.zeros_like(P[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(P[k]) for k in self.keys}

    def step(self, images, labels):
        P = self.P
        leaves = {k: P[k].detach().requires_grad_(True) for k in self.keys}
        Q = dict(P)
        Q.update(leaves)
        out, losses = slot_model_forward(Q, images, labels, self.cfg, training=True)
        grads = torch.autograd.grad(losses[0], [leaves[k] for k in self.keys], allow_unused=True)
        for k in P:  # batch-norm buffers were updated in place inside Q (same tensor objects)
            if k not in leaves:
                P[k] = Q[k]
        self.t += 1
        with torch.no_grad():
            for k, g in zip(self.keys, grads):
                if g is None:
                    continue
                adamw_step(P[k], g, self.m[k], self.v[k], self.t, lr=self.lr)
        acc = (out.argmax(1) == labels).sum().float().item() / labels.size(0)   # calculate_tool.py:4-7
        return out.detach(), [float(l) for l in losses], acc, dict(zip(self.keys, grads))


# --------------------------------------------------------------------------------------------------------------
# deterministic parameter / input synthesis shared by the golden generator, the tests and bench.py
# --------------------------------------------------------------------------------------------------------------
def state_dict_spec(arch, num_classes, slots_per_class, to_k_layer, in_chans=3, hidden_dim=64, mnist_stem=False,
                    use_slot=True):
    """Ordered {key: shape} of SlotModel.state_dict() for the three supported backbones.
    Cross-checked against the reference's own state_dict in tests/test_oracle_vs_reference.py."""
    spec = {}

    def bn(name, c):
        spec[name + ".weight"] = (c,)
        spec[name + ".bias"] = (c,)
        spec[name + ".running_mean"] = (c,)
        spec[name + ".running_var"] = (c,)
        spec[name + ".num_batches_tracked"] = ()

    cfg = ARCHS[arch]
    pre = "backbone."
    if cfg["kind"] == "resnest":
        spec[pre + "conv1.0.weight"] = (32, in_chans, 3, 3)
        bn(pre + "conv1.1", 32)
        spec[pre + "conv1.3.weight"] = (32, 32, 3, 3)
        bn(pre + "conv1.4", 32)
        spec[pre + "conv1.6.weight"] = (64, 32, 3, 3)
    elif mnist_stem:
        spec[pre + "conv1.weight"] = (64, 1, 3, 3)
    else:
        spec[pre + "conv1.weight"] = (64, in_chans, 7, 7)
    bn(pre + "bn1", 64)
    inplanes = 64
    for li, nblocks in enumerate(cfg["layers"]):
        planes = 64 * 2 ** li
        for bi in range(nblocks):
            name = f"{pre}layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            if cfg["kind"] == "resnest":
                gw = planes
                spec[name + ".conv1.weight"] = (gw, inplanes, 1, 1)
                bn(name + ".bn1", gw)
                attn = max(gw * 2 // 4, 32)
                spec[name + ".conv2.conv.weight"] = (gw * 2, gw // 2, 3, 3)
                bn(name + ".conv2.bn0", gw * 2)
                spec[name + ".conv2.fc1.weight"] = (attn, gw, 1, 1)
                spec[name + ".conv2.fc1.bias"] = (attn,)
                bn(name + ".conv2.bn1", attn)
                spec[name + ".conv2.fc2.weight"] = (gw * 2, attn, 1, 1)
                spec[name + ".conv2.fc2.bias"] = (gw * 2,)
                spec[name + ".conv3.weight"] = (planes * 4, gw, 1, 1)
                bn(name + ".bn3", planes * 4)
                if bi == 0:
                    spec[name + ".downsample.1.weight"] = (planes * 4, inplanes, 1, 1)
                    bn(name + ".downsample.2", planes * 4)
                    inplanes = planes * 4
            else:
                spec[name + ".conv1.weight"] = (planes, inplanes, 3, 3)
                bn(name + ".bn1", planes)
                spec[name + ".conv2.weight"] = (planes, planes, 3, 3)
                bn(name + ".bn2", planes)
                if bi == 0 and (stride != 1 or inplanes != planes):
                    spec[name + ".downsample.0.weight"] = (planes, inplanes, 1, 1)
                    bn(name + ".downsample.1", planes)
                inplanes = planes
    if not use_slot:
        spec[pre + "fc.weight"] = (num_classes, cfg["channel"])
        spec[pre + "fc.bias"] = (num_classes,)
        return spec
    d = hidden_dim
    spec["conv1x1.weight"] = (d, cfg["channel"], 1, 1)
    spec["conv1x1.bias"] = (d,)
    spec["slot.initial_slots"] = (1, num_classes * slots_per_class, d)
    spec["slot.to_q.0.weight"] = (d, d)
    spec["slot.to_q.0.bias"] = (d,)
    for l in range(to_k_layer):
        spec[f"slot.to_k.{2 * l}.weight"] = (d, d)
        spec[f"slot.to_k.{2 * l}.bias"] = (d,)
    spec["slot.gru.weight_ih_l0"] = (3 * d, d)
    spec["slot.gru.weight_hh_l0"] = (3 * d, d)
    spec["slot.gru.bias_ih_l0"] = (3 * d,)
    spec["slot.gru.bias_hh_l0"] = (3 * d,)
    return spec


def synth_value(key, shape, rng):
    """Seeded synthetic value for one state_dict entry (numpy fp32/int64).  Chosen so that every layer is
    exercised (non-zero last-BN gamma) and activations stay O(1) through 50 layers."""
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if key.endswith("running_mean"):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    leaf = key.rsplit(".", 2)[-2] if key.count(".") >= 2 else ""
    is_bn = (leaf.startswith("bn") or (leaf.isdigit() and len(shape) == 1)) and "slot." not in key \
        and "conv1x1" not in key and ".fc" not in key
    if is_bn and key.endswith(".weight"):
        if leaf == "bn3" or leaf == "bn2" and "conv2" not in key:
            return rng.uniform(0.3, 0.7, shape).astype(np.float32)      # residual-branch last BN
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if is_bn and key.endswith(".bias"):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if key == "slot.initial_slots":
        d = shape[-1]
        mu = rng.standard_normal((1, 1, d))
        sigma = np.abs(rng.standard_normal((1, 1, d)))
        return (mu + sigma * rng.standard_normal(shape)).astype(np.float32)     # slot_attention.py:20-25
    if key.startswith("slot.") or key.endswith(".bias"):
        fan_in = shape[-1] if len(shape) > 1 else 64
        bound = 1.0 / math.sqrt(fan_in)
        return rng.uniform(-bound, bound, shape).astype(np.float32)             # nn.Linear / nn.GRU default
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return (rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)
    if len(shape) == 2:
        return (rng.standard_normal(shape) * math.sqrt(1.0 / shape[1])).astype(np.float32)
    raise ValueError(f"no synthesis rule for {key} {shape}")


def synth_state(spec, seed):
    """{key: torch tensor} for a spec, values drawn key-by-key from default_rng(seed) in spec order."""
    rng = np.random.default_rng(seed)
    return {k: torch.from_numpy(np.asarray(synth_value(k, shp, rng))) for k, shp in spec.items()}


def synth_batch(B, in_chans, H, num_classes, seed):
    rng = np.random.default_rng(seed)
    images = torch.from_numpy(rng.standard_normal((B, in_chans, H, H), dtype=np.float32))
    labels = torch.from_numpy(rng.integers(0, num_classes, B).astype(np.int64))
    return images, labels
