"""ResNeSt bottleneck (radix 2, cardinality 1, avd after the split-attention conv, avg-pool downsample).
Mirrors timm/models/resnest.py:58-143 and :161-189 of the reference."""
import os

import torch
import torch.nn as nn

from ... import kernels as K
from ...nn_hip import Act, BatchNorm2d, Conv2d
from .layers.split_attn import SplitAttnConv2d
from .resnet import AvgPool2dSpec, ResNet, BRANCH_FWD, BRANCH_BWD


# bf16 storage of the masked block-output gradients (the residual-stream gradient) next to the bf16-stored activations;
# SCOUTER_BF16_GRAD_STREAM=0: fp32.  Unlike Conv2d.grad_storage this one rounds values that fp32 arithmetic reads (the
# BatchNorm backward, the shortcut addend): the oracle emulates it (oracle.torch_oracle.ACTIVATION_STORAGE).
GRAD_STREAM_BF16_DEFAULT = os.environ.get("SCOUTER_BF16_GRAD_STREAM", "1") != "0"


class ResNestBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, radix=2, avd=True, avd_first=False, **_):
        super().__init__()
        if avd_first or not avd:
            raise NotImplementedError("only the avd / avd_last configuration of resnest26d/50d is supported")
        gw = planes
        avd_stride = stride if stride > 1 else 0          # `is_first` is never set by ResNet._make_layer
        self.conv1 = Conv2d(inplanes, gw, 1)
        self.bn1 = BatchNorm2d(gw)
        self.act1 = Act()
        self.conv2 = SplitAttnConv2d(gw, gw, 3, 1, 1, radix=radix)
        self.avd_last = AvgPool2dSpec(3, avd_stride, 1) if avd_stride > 0 else None
        self.conv3 = Conv2d(gw, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.act3 = Act()
        self.downsample = downsample
        # bf16 activation storage (SlotModel.set_precision("bf16"); never the network's last block, whose output feeds the
        # fp32 head): conv3 / downsample outputs and the block output -- the 4x-wide tensors -- are stored as bf16
        self.store_bf16 = False
        # ... and the masked block-INPUT gradient this block's conv1 produces (the previous block's output gradient): a
        # per-block attribute set together with store_bf16 by SlotModel.set_activation_storage (no module global)
        self.grad_stream_bf16 = False

    def _storage(self, x):
        """Storage type of this block's wide tensors for an input batch x: bf16 where the bf16-input kernels run
        (their pixel-count rule), fp32 otherwise.  The next block's conv1 sees the same pixel count, so a bf16-stored
        output is always read by a bf16-input kernel."""
        if not (self.store_bf16 and self.conv3.precision == "bf16"):
            return None
        B, H, W, _ = x.shape
        if self.avd_last is not None:
            H, W = K.pool_out(H, 3, self.avd_last.s, 1), K.pool_out(W, 3, self.avd_last.s, 1)
        return K.BF16 if B * H * W >= K.BF16_MIN_PIXELS else None

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn3.weight)

    def fwd(self, x, save, tracked):
        res, rbn, kd = (x, None, None)
        odt = self._storage(x)
        if self.downsample is not None:       # (on its own stream next to the main branch, Downsample.fwd_async)
            res, rbn, kd, hnd = self.downsample.fwd_async(x, save, tracked, BRANCH_FWD and self.conv1.use_side_stream,
                                                          out_dtype=odt)
        c1, k1 = self.conv1.fwd(x, save, bn_stats=self.bn1.training)
        t1 = c1[0] if isinstance(c1, tuple) else c1
        conv = self.conv2.conv                   # (all three of its kernels on planes: no fp32 copy of h1 is written)
        h1, b1 = self.bn1.fwd(c1, save, relu=True, tracked=tracked, planes=conv.planes_in(),
                              keep_f32=not conv.planes_only(t1.shape[1], t1.shape[2], t1.shape[0]))
        # (stored blocks: the radix convolution's output -- the tensor the split attention reads four times -- as bf16; the
        #  attention-weighted sum too where conv3 alone reads it: the bf16-input kernels round it the same way)
        x0dt = K.BF16 if (self.store_bf16 and conv.precision == "bf16" and
                          t1.shape[0] * t1.shape[1] * t1.shape[2] >= K.BF16_MIN_PIXELS) else None
        sadt = self.conv3.act_storage(*t1.shape[:3]) if self.avd_last is None else None
        sa, ksa = self.conv2.fwd(h1, save, tracked, x0_dtype=x0dt, out_dtype=sadt)
        p = self.avd_last.fwd(sa) if self.avd_last is not None else sa
        c3, k3 = self.conv3.fwd(p, save, bn_stats=self.bn3.training, out_dtype=odt)
        if self.downsample is not None:
            self.downsample.fwd_join(x.device, hnd)
        out, b3 = self.bn3.fwd(c3, save, relu=True, residual=res, tracked=tracked, residual_bn=rbn, out_dtype=odt)
        return out, ((k1, b1, ksa, tuple(sa.shape), k3, b3, kd) if save else None)

    def out_fuse(self, ctx):
        """What the kernel producing this block's OUTPUT gradient needs to finish the backward reductions of bn3 (and
        of the downsample BatchNorm, fed by the same masked gradient) in its epilogue."""
        b3, kd = ctx[5], ctx[6]
        return BatchNorm2d.fuse(b3, *([kd[1]] if self.downsample is not None else []))

    def bwd(self, dout, ctx, need_dx=True, own=None, post=None):
        """own: this block's out_fuse object if the producer of `dout` applied it (dout is then the masked gradient and
        the bn3 / downsample sums are reduced); post: the previous block's, for conv1's input-gradient epilogue."""
        k1, b1, ksa, sa_shape, k3, b3, kd = ctx
        own = own if own is not None and own.applied else None
        # (dc3 / dc1 / the downsample gradient are read by bf16-input kernels only: stored as bf16 where those run)
        dc3, dres = self.bn3.bwd(dout, b3, want_gout=True, fused=own.ext(0) if own else None,
                                 dx_dtype=self.conv3.grad_storage(*dout.shape[:3]))
        dxres = dres
        if self.downsample is not None:       # the branch's backward next to the main branch's (Downsample.bwd_async)
            dxres, hnd = self.downsample.bwd_async(dres, kd, need_dx, own.ext(1) if own else None,
                                                   BRANCH_BWD and self.conv1.use_side_stream)
        dp = self.conv3.bwd(dc3, k3, True)
        dsa = self.avd_last.bwd(dp, sa_shape) if self.avd_last is not None else dp
        f1 = BatchNorm2d.fuse(b1)
        dh1 = self.conv2.bwd(dsa, ksa, post=f1)
        dc1, _ = self.bn1.bwd(dh1, b1, fused=f1.ext(0) if f1.applied else None,
                              dx_dtype=self.conv1.grad_storage(*dh1.shape[:3]))
        if self.downsample is not None:
            dxres = self.downsample.bwd_join(dxres, dres.device, hnd)
        # the block-INPUT gradient this produces is the previous block's masked output gradient (post): stored as bf16 where
        # that block stores its activations as bf16 (same rule: ResNestBottleneck._storage of the previous block)
        gdt = K.BF16 if (post is not None and need_dx and self.grad_stream_bf16 and isinstance(k1, torch.Tensor) and
                         k1.dtype == K.BF16) else None
        return self.conv1.bwd(dc1, k1, need_dx, addend=dxres, post=post if need_dx else None, dx_dtype=gdt)


def _resnest(name, layers, pretrained, num_classes, in_chans, **kwargs):
    model = ResNet(ResNestBottleneck, layers, num_classes=num_classes, in_chans=in_chans, stem_type="deep",
                   stem_width=32, avg_down=True, block_args=dict(radix=2, avd=True, avd_first=False), **kwargs)
    if pretrained:                               # resnest.py:165-189; local file instead of a download (helpers.py)
        from .helpers import load_pretrained
        load_pretrained(model, name, num_classes, in_chans)
    return model


def resnest26d(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _resnest("resnest26d", [2, 2, 2, 2], pretrained, num_classes, in_chans, **kwargs)


def resnest50d(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _resnest("resnest50d", [3, 4, 6, 3], pretrained, num_classes, in_chans, **kwargs)
