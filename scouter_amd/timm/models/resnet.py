"""ResNet trunk + BasicBlock on the HIP layers.  Mirrors timm/models/resnet.py:134-199 (BasicBlock), :273-306
(downsample_conv / downsample_avg), :380-509 (ResNet) of the reference for the configurations the xSlot path
uses: stem '' (7x7, or the MNIST 3x3 1-channel stem swapped in by sloter/slot_model.py:23-24) and 'deep'
(32-32-64), output stride 32, no drop-path / drop-block / anti-aliasing.  Activations are NHWC internally."""
import os

import torch
import torch.nn as nn

from ...nn_hip import Act, BatchNorm2d, Conv2d, LinearParams, StemConv2d
from ... import kernels as K

BRANCH_FWD = os.environ.get("SCOUTER_SIDE_FWD", "1") == "1"      # shortcut branch of a downsampling block on its own stream
BRANCH_BWD = os.environ.get("SCOUTER_SIDE_BWD", "1") == "1"


class Identity(nn.Module):
    def forward(self, x):
        return x


class AvgPool2dSpec(nn.Module):
    """Holds the pooling geometry of nn.AvgPool2d (the op itself runs in scouter_avgpool_*_f32)."""

    def __init__(self, kernel_size, stride, padding=0, ceil_mode=False, count_include_pad=True):
        super().__init__()
        self.k, self.s, self.p, self.ceil, self.cip = kernel_size, stride, padding, ceil_mode, count_include_pad

    def fwd(self, x):
        return K.avgpool_fwd(x, self.k, self.s, self.p, self.ceil, self.cip)

    def bwd(self, dy, x_shape):
        return K.avgpool_bwd(dy, x_shape, self.k, self.s, self.p, self.ceil, self.cip)


def _map_of(c):
    """(B, H, W) of a convolution result that may come as (tensor, statistics)"""
    t = c[0] if isinstance(c, tuple) else c
    return tuple(t.shape[:3])


class Downsample(nn.Sequential):
    """downsample_conv: [conv1x1(stride), bn]   downsample_avg: [pool | Identity, conv1x1, bn]"""

    def fwd(self, x, save, tracked, out_dtype=None):
        """-> (raw convolution output, saved block of the BatchNorm, ctx): the BatchNorm itself is applied by the
        block's last BatchNorm pass together with its own (BatchNorm2d.fwd residual_bn) -- the normalised shortcut is
        never stored.  ctx[1] has the layout of a BatchNorm2d context (x, mask = None, saved, training).
        out_dtype: storage type of the raw output (bf16 activation storage, ResNestBottleneck.fwd)."""
        mods = list(self)
        pool = None
        if len(mods) == 3:
            pool = mods[0] if isinstance(mods[0], AvgPool2dSpec) else None
            mods = mods[1:]
        xin = pool.fwd(x) if pool is not None else x
        c, c_conv = mods[0].fwd(xin, save, bn_stats=mods[1].training, out_dtype=out_dtype)
        craw, saved = mods[1].stats_only(c, tracked, relu_follows=False)
        c_bn = (craw, None, saved, mods[1].training)
        return craw, saved, ((c_conv, c_bn, pool, tuple(x.shape)) if save else None)

    # ---- the branch on its own stream ("branch": never behind the backlog of weight gradients), next to the block's
    # main branch whose many small launches leave the GPU idle in between.  Measured on the benchmark step (3 x 120 steps,
    # interleaved, one box): forward +0.5 % (3 970 -> 3 991), backward +1.6 % (3 947 -> 4 011).
    def fwd_async(self, x, save, tracked, on, out_dtype=None):
        """-> (raw output, saved block, ctx, handle); handle goes to fwd_join (no per-call state on the module)."""
        if not on:
            return self.fwd(x, save, tracked, out_dtype) + (None,)
        with K.side_stream(x.device, x, enabled=True, which="branch"):
            craw, saved, ctx = self.fwd(x, save, tracked, out_dtype)
        made = [craw, saved] + ([ctx[0][0] if isinstance(ctx[0], tuple) else ctx[0]] if ctx is not None else [])
        return craw, saved, ctx, [t for t in made if torch.is_tensor(t) and t is not x]

    @staticmethod
    def fwd_join(device, handle):
        if handle is not None:
            K.join_side_stream(device, "branch")
            cur = torch.cuda.current_stream(device)
            for t in handle:
                t.record_stream(cur)

    def bwd_async(self, dy, ctx, need_dx, fused, on):
        """-> (dx, handle) for bwd_join."""
        if not on:
            return self.bwd(dy, ctx, need_dx, fused=fused), False
        with K.side_stream(dy.device, dy, fused[0] if fused else None, enabled=True, which="branch"):
            return self.bwd(dy, ctx, need_dx, fused=fused), True

    @staticmethod
    def bwd_join(dx, device, handle):
        if handle:
            K.join_side_stream(device, "branch")
            if dx is not None:
                dx.record_stream(torch.cuda.current_stream(device))
        return dx

    def bwd(self, dy, ctx, need_dx, fused=None):
        c_conv, c_bn, pool, x_shape = ctx
        mods = list(self)[-2:]
        dc, _ = mods[1].bwd(dy, c_bn, fused=fused, dx_dtype=mods[0].grad_storage(*dy.shape[:3]))
        dx = mods[0].bwd(dc, c_conv, need_dx)
        if dx is not None and pool is not None:
            dx = pool.bwd(dx, x_shape)
        return dx


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, **_):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = BatchNorm2d(planes)
        self.act1 = Act()
        self.conv2 = Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = BatchNorm2d(planes)
        self.act2 = Act()
        self.downsample = downsample

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn2.weight)

    def fwd(self, x, save, tracked):
        res, rbn, kd = (x, None, None)
        if self.downsample is not None:
            # (not on the branch stream: resnet18's step is launch-bound -- measured 7 460 -> 7 280 img/s with it)
            res, rbn, kd, hnd = self.downsample.fwd_async(x, save, tracked, False)
        c1, k1 = self.conv1.fwd(x, save, bn_stats=self.bn1.training)
        h1, b1 = self.bn1.fwd(c1, save, relu=True, tracked=tracked)
        c2, k2 = self.conv2.fwd(h1, save, bn_stats=self.bn2.training)
        if self.downsample is not None:
            self.downsample.fwd_join(x.device, hnd)
        out, b2 = self.bn2.fwd(c2, save, relu=True, residual=res, tracked=tracked, residual_bn=rbn)
        return out, ((k1, b1, k2, b2, kd) if save else None)

    def out_fuse(self, ctx):
        """see ResNestBottleneck.out_fuse: bn2 (+ the downsample BatchNorm)"""
        b2, kd = ctx[3], ctx[4]
        return BatchNorm2d.fuse(b2, *([kd[1]] if self.downsample is not None else []))

    def bwd(self, dout, ctx, need_dx=True, own=None, post=None):
        k1, b1, k2, b2, kd = ctx
        own = own if own is not None and own.applied else None
        dc2, dres = self.bn2.bwd(dout, b2, want_gout=True, fused=own.ext(0) if own else None)
        dxres = dres
        if self.downsample is not None:
            dxres, hnd = self.downsample.bwd_async(dres, kd, need_dx, own.ext(1) if own else None, False)
        f1 = BatchNorm2d.fuse(b1)
        dh1 = self.conv2.bwd(dc2, k2, True, post=f1)
        dc1, _ = self.bn1.bwd(dh1, b1, fused=f1.ext(0) if f1.applied else None)
        if self.downsample is not None:
            dxres = self.downsample.bwd_join(dxres, dres.device, hnd)
        return self.conv1.bwd(dc1, k1, need_dx, addend=dxres, post=post if need_dx else None)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, in_chans=3, stem_width=64, stem_type="", avg_down=False,
                 zero_init_last_bn=True, block_args=None, **_):
        super().__init__()
        block_args = block_args or {}
        self.num_classes = num_classes
        deep_stem = "deep" in stem_type
        self.inplanes = stem_width * 2 if deep_stem else 64
        if deep_stem:
            self.conv1 = nn.Sequential(
                StemConv2d(in_chans, stem_width, 3, 2, 1), BatchNorm2d(stem_width), Act(),
                Conv2d(stem_width, stem_width, 3, 1, 1), BatchNorm2d(stem_width), Act(),
                Conv2d(stem_width, self.inplanes, 3, 1, 1))
        else:
            self.conv1 = StemConv2d(in_chans, self.inplanes, 7, 2, 3)
        self.bn1 = BatchNorm2d(self.inplanes)
        self.act1 = Act()
        self.maxpool = Identity()      # MaxPool2d(3, 2, 1) runs in scouter_maxpool_*_f32
        self._capture = None           # test instrumentation: (dict, key) -> the max-pool window indices go there
        # called after the first convolution has been queued, right before the first BatchNorm reads / updates its running
        # statistics (scouter_amd.parallel: the compute stream waits for the asynchronous DDP buffer broadcast HERE)
        self._pre_bn_hooks = []
        # called after the stem (max-pool queued), before layer1: SlotModel joins the stream that split this step's weight
        # planes next to the stem's kernels
        self._post_stem_hooks = []
        self.fuse_stem_pool = os.environ.get("SCOUTER_FUSE_STEM_POOL", "1") == "1"   # bn1 + act1 + maxpool in one pass
        chans, strides = [64, 128, 256, 512], [1, 2, 2, 2]
        for i in range(4):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, chans[i], layers[i], strides[i], avg_down,
                                                                block_args))
        self.num_features = 512 * block.expansion
        self.global_pool = Identity()
        self.fc = LinearParams(self.num_features, num_classes)
        if zero_init_last_bn:                       # resnet.py:455-458
            for m in self.modules():
                if hasattr(m, "zero_init_last_bn"):
                    m.zero_init_last_bn()

    def _make_layer(self, block, planes, blocks, stride, avg_down, block_args):
        downsample = None
        outp = planes * block.expansion
        if stride != 1 or self.inplanes != outp:
            if avg_down:
                pool = AvgPool2dSpec(2, stride, ceil_mode=True, count_include_pad=False) if stride != 1 else Identity()
                downsample = Downsample(pool, Conv2d(self.inplanes, outp, 1), BatchNorm2d(outp))
            else:
                downsample = Downsample(Conv2d(self.inplanes, outp, 1, stride, 0), BatchNorm2d(outp))
        mods = [block(self.inplanes, planes, stride, downsample, **block_args)]
        self.inplanes = outp
        mods += [block(self.inplanes, planes, **block_args) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    # ---- explicit forward / backward over NHWC tensors
    def features_fwd(self, x_nchw, save, tracked=None):
        ctx = []
        joined = [False]

        def join_weight_planes():
            # the stream that splits this step's weight planes is joined ONCE per forward, in front of the first kernel that
            # reads planes: layer1 by default, a stem convolution when it runs on the register-split kernel (SCOUTER_X3 bit 4)
            if not joined[0]:
                joined[0] = True
                for hook in self._post_stem_hooks:
                    hook()
        if isinstance(self.conv1, nn.Sequential):
            s = self.conv1
            # (bf16 mode: the two stem activations are read by bf16-input kernels only -- stored as bf16, same results)
            c, k0 = s[0].fwd(x_nchw, save, bn_stats=s[1].training)
            for hook in self._pre_bn_hooks:
                hook()
            h, b0 = s[1].fwd(c, save, relu=True, tracked=tracked, out_dtype=s[3].act_storage(*_map_of(c)))
            if s[3].x3_mode():
                join_weight_planes()
            c, k1 = s[3].fwd(h, save, bn_stats=s[4].training)
            h, b1 = s[4].fwd(c, save, relu=True, tracked=tracked, out_dtype=s[6].act_storage(*_map_of(c)))
            if s[6].x3_mode():
                join_weight_planes()
            c, k2 = s[6].fwd(h, save, bn_stats=self.bn1.training)
            ctx.append((k0, b0, k1, b1, k2))
        else:
            c, k0 = self.conv1.fwd(x_nchw, save, bn_stats=self.bn1.training)
            for hook in self._pre_bn_hooks:
                hook()
            ctx.append((k0,))
        # bn1 + act1 + maxpool in one pass: only bn1's statistics are finalised, relu(bn1(c)) is evaluated inside the
        # pooling windows (and again, with the same fma, in the backward) -- the 112x112 activation is never stored
        ct = c[0] if isinstance(c, tuple) else c
        # (the fused backward walks one workgroup row per input pixel row: grid.y = B * H <= 65535, i.e. 585 images of
        # 224 x 224 per GPU; larger batches take the unfused chain, whose pool backward has a generic fallback)
        if self.fuse_stem_pool and ct.shape[0] * ct.shape[1] <= 65535:
            craw, saved = self.bn1.stats_only(c, tracked)
            p, arg = K.bn_maxpool_fwd(craw, saved, 3, 2, 1, want_argmax=save)
            ctx.append((craw, saved, self.bn1.training, arg))
        else:
            h, bb = self.bn1.fwd(c, save, relu=True, tracked=tracked)
            p, arg = K.maxpool_fwd(h, 3, 2, 1, want_argmax=save)
            ctx.append((bb, arg, tuple(h.shape)))
        if self._capture is not None and arg is not None:
            self._capture[0][self._capture[1]] = arg
        x = p
        join_weight_planes()
        for li in range(1, 5):
            for blk in getattr(self, "layer%d" % li):
                x, c_blk = blk.fwd(x, save, tracked)
                ctx.append(c_blk)
        return x, (ctx if save else None)

    def _first_trainable_stage(self):
        """Index of the earliest stage (0 = stem, 1..4 = layer1..4) holding a trainable parameter; 5 if none."""
        stages = [[self.conv1, self.bn1]] + [[getattr(self, "layer%d" % i)] for i in range(1, 5)]
        for i, mods in enumerate(stages):
            if any(p.requires_grad for m in mods for p in m.parameters()):
                return i
        return 5

    def last_fuse(self, ctx):
        """BnBwdFuse of the LAST block's output BatchNorm(s), for whatever kernel produces d(features) (the head's
        conv1x1 input gradient); None when the backbone gets no gradient."""
        if self._first_trainable_stage() > 4:
            return None
        return self.layer4[-1].out_fuse(ctx[-1])

    def features_bwd(self, dfeat, ctx, on_stage_done=None, own=None):
        """`on_stage_done(name)` is called when every gradient of layer4 / layer3 / layer2 / layer1 has been
        written (used to launch that stage's gradient all-reduce while earlier layers are still in backward).
        BatchNorm-backward reductions ride in the epilogue of the input-gradient kernel that produces each gradient
        (K.BnBwdFuse): block i's conv1 finishes block i-1's bn3 / downsample sums, `own` is the last block's object if
        the producer of `dfeat` applied it."""
        first = self._first_trainable_stage()
        blocks = [(li, blk) for li in range(1, 5) for blk in getattr(self, "layer%d" % li)]
        d = dfeat
        for idx in range(len(blocks) - 1, -1, -1):
            li, blk = blocks[idx]
            if li < first:
                return
            is_first_trainable = li == first and (idx == 0 or blocks[idx - 1][0] < first)
            post = blocks[idx - 1][1].out_fuse(ctx[2 + idx - 1]) if idx > 0 and not is_first_trainable else None
            d = blk.bwd(d, ctx[2 + idx], need_dx=not is_first_trainable, own=own, post=post)
            own = post
            if on_stage_done is not None and (idx == 0 or blocks[idx - 1][0] != li):
                on_stage_done("layer%d" % li)
        if first > 0:
            return
        if len(ctx[1]) == 4:
            craw, saved, training1, arg = ctx[1]
            last = self.conv1[6] if isinstance(self.conv1, nn.Sequential) else None
            dc = K.bn_maxpool_bwd(d, arg, craw, saved, training1, self.bn1._dg, self.bn1._db, 3, 2, 1,
                                  dx_dtype=(last.grad_storage(*craw.shape[:3]) if last is not None else None) or K.F32)
        else:
            bb, arg, hshape = ctx[1]
            dh = K.maxpool_bwd(d, arg, hshape, 3, 2, 1)
            dc, _ = self.bn1.bwd(dh, bb)
        if isinstance(self.conv1, nn.Sequential):
            s = self.conv1
            k0, b0, k1, b1, k2 = ctx[0]
            f = BatchNorm2d.fuse(b1)
            dh = s[6].bwd(dc, k2, True, post=f)
            dc, _ = s[4].bwd(dh, b1, fused=f.ext(0) if f.applied else None, dx_dtype=s[3].grad_storage(*dh.shape[:3]))
            f = BatchNorm2d.fuse(b0)
            dh = s[3].bwd(dc, k1, True, post=f)
            dc, _ = s[1].bwd(dh, b0, fused=f.ext(0) if f.applied else None)
            s[0].bwd(dc, k0, False)
        else:
            self.conv1.bwd(dc, ctx[0][0], False)

    @torch.no_grad()
    def forward_features(self, x):
        """NCHW in -> NCHW out (inference helper; training goes through SlotModel's fused autograd node)."""
        feat, _ = self.features_fwd(x.float().contiguous(), False)
        return K.nhwc_to_nchw(feat)

    # ---- classifier head of the FC baseline: global average pool + Linear (resnet.py:503-509)
    def classifier_fwd(self, feat, save):
        B, H, W, C = feat.shape
        pooled = K.avgpool_fwd(feat, H, H, 0, False, True).view(B, C)
        logits = K.linear_small_fwd(pooled, self.fc.weight, self.fc.bias)
        return logits, ((pooled, (B, H, W, C)) if save else None)

    def classifier_bwd(self, dlogits, ctx, need_dfeat):
        pooled, (B, H, W, C) = ctx
        dpooled = K.linear_small_bwd(dlogits, pooled, self.fc.weight, self.fc._g.get("weight"), self.fc._g.get("bias"),
                                     need_dx=need_dfeat)
        if not need_dfeat:
            return None
        return K.avgpool_bwd(dpooled.view(B, 1, 1, C), (B, H, W, C), H, H, 0, False, True)

    @torch.no_grad()
    def forward(self, x):
        feat, _ = self.features_fwd(x.float().contiguous(), False)
        if isinstance(self.fc, LinearParams):
            return self.classifier_fwd(feat, False)[0]
        return K.nhwc_to_nchw(feat).flatten(1)    # global_pool / fc replaced by Identical (slot_model.py:38-40)


def resnet18(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    model = ResNet(BasicBlock, [2, 2, 2, 2], num_classes=num_classes, in_chans=in_chans, **kwargs)
    if pretrained:                               # resnet.py:516-521; local file instead of a download (helpers.py)
        from .helpers import load_pretrained
        load_pretrained(model, "resnet18", num_classes, in_chans)
    return model
