"""Split-attention convolution (ResNeSt), radix 2 / cardinality 1 -- HIP forward + hand-written backward.
Mirrors timm/models/layers/split_attn.py:31-80 of the reference (module / parameter names: conv, bn0, fc1, bn1, fc2).
Channel layout of the radix conv output: [radix 0: C' channels | radix 1: C' channels] (split_attn.py:64-66)."""
import os

import torch.nn as nn

from ....nn_hip import Act, BatchNorm2d, Conv2d
from .... import kernels as K

# the d(attention) pass also reduces the per-image statistics of bn0's backward (kernels.sa_dattn want_stats)
SA_SUMS = os.environ.get("SCOUTER_SA_SUMS", "1") == "1"


class SplitAttnConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, groups=1, radix=2,
                 reduction_factor=4):
        super().__init__()
        if radix != 2 or groups != 1:
            raise NotImplementedError("scouter_amd supports the radix-2 / cardinality-1 split attention of "
                                      "resnest26d / resnest50d only")
        self.radix = radix
        mid_chs = out_channels * radix
        attn_chs = max(in_channels * radix // reduction_factor, 32)
        self.conv = Conv2d(in_channels, mid_chs, kernel_size, stride, padding, groups=groups * radix, bias=False)
        self.bn0 = BatchNorm2d(mid_chs)
        self.act0 = Act()
        self.fc1 = Conv2d(out_channels, attn_chs, 1, bias=True)
        self.bn1 = BatchNorm2d(attn_chs)
        self.act1 = Act()
        self.fc2 = Conv2d(attn_chs, mid_chs, 1, bias=True)

    def fwd(self, x, save, tracked=None, x0_dtype=None, out_dtype=None):
        """bn0 + ReLU are folded into the two passes that read the radix convolution's output anyway (GAP and the
        attention-weighted sum): only the batch statistics are finalised, the activation relu(bn0(x0)) is never stored --
        one 8-byte-per-element pass less on the largest tensor of every block, forward and backward.
        x0_dtype / out_dtype = torch.bfloat16 (bf16 activation storage): the raw convolution output x0 -- read four times,
        twice in each direction -- and the attention-weighted sum are stored as bf16."""
        c, c_conv = self.conv.fwd(x, save, bn_stats=self.bn0.training, out_dtype=x0_dtype)
        x0, saved0 = self.bn0.stats_only(c, tracked)                             # x0: [B,H,W,2C'] raw conv output
        B = x0.shape[0]
        gap = K.sa_gap(x0, saved0)                                               # split_attn.py:63-68 (+ bn0, act0)
        z1, c_fc1 = self.fc1.fwd(gap.view(B, 1, 1, -1), save, bn_stats=self.bn1.training)
        g1, c_bn1 = self.bn1.fwd(z1, save, relu=True, tracked=tracked)
        z2, c_fc2 = self.fc2.fwd(g1, save)
        a = K.radix_softmax_fwd(z2.view(B, -1))                                  # split_attn.py:20-28,75
        out = K.sa_apply_fwd(x0, a, saved0, out_dtype=out_dtype or K.F32)        # split_attn.py:76-79
        return out, ((c_conv, x0, saved0, self.bn0.training, c_fc1, c_bn1, c_fc2, a) if save else None)

    def bwd(self, dout, ctx, post=None):
        """post: K.BnBwdFuse of the BatchNorm in front of this layer (finished in the input-gradient epilogue)."""
        c_conv, x0, saved0, training0, c_fc1, c_bn1, c_fc2, a = ctx
        B = x0.shape[0]
        da, sums = K.sa_dattn(x0, dout, saved0, want_stats=True) if SA_SUMS else (K.sa_dattn(x0, dout, saved0), None)
        dz2 = K.radix_softmax_bwd(a, da)
        dg1 = self.fc2.bwd(dz2.view(B, 1, 1, -1), c_fc2, True)
        dz1, _ = self.bn1.bwd(dg1, c_bn1)
        dgap = self.fc1.bwd(dz1, c_fc1, True)
        on_planes = isinstance(c_conv, tuple) and c_conv[1] is not None
        only = on_planes and c_conv[2] is not None and self.conv.planes_only(x0.shape[1], x0.shape[2])
        planes = self.conv.planes_dy() if on_planes else 0
        if not planes and self.conv.dy_plane_only(*x0.shape[:3]):
            planes, only = 1, True       # (bf16 mode, 32-channel groups: the gradient as ONE bf16 plane for the igemm kernels)
        dc = K.sa_bn_bwd(dout, a, dgap.view(B, -1), x0, saved0, training0, self.bn0._dg, self.bn0._db,
                         planes=planes, keep_f32=not only, sums=sums)
        return self.conv.bwd(dc, c_conv, True, post=post)
