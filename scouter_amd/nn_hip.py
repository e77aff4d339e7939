"""Layer primitives with EXPLICIT forward / backward over NHWC device tensors (the host side of the HIP path).

The reference leaves sequencing of the backward pass to torch.autograd over ~1400 tiny ATen kernels
(SURVEY.md section 2.2).  Here every layer has `fwd(x, save, ...) -> (y, ctx)` and `bwd(dy, ctx, ...) -> dx`
that launch the hand-written kernels directly; parameter gradients are written straight into one flat fp32
arena (GradArena) whose slices are exposed as `param.grad`, so the data-parallel all-reduce and the fused AdamW
see a single contiguous buffer.  Modules keep the reference's parameter / buffer names so state_dicts are
interchangeable; conv weights have the reference's logical (Cout, Cin/g, kh, kw) shape
over an HWIO physical layout."""
import math
import os

import torch
import torch.nn as nn

from . import kernels as K


class Act(nn.Module):
    """Placeholder for the reference's ReLU modules (keeps nn.Sequential indices, e.g. backbone.conv1.{0,1,3,4,6})."""

    def forward(self, x):   # pragma: no cover - activation is fused into the neighbouring kernel
        raise RuntimeError("activation is fused; call the parent module")


# bf16 storage of the gradients that only bf16-input convolution kernels read (Conv2d.grad_storage); SCOUTER_BF16_GRADS=0: fp32
GRAD_STORAGE_BF16 = os.environ.get("SCOUTER_BF16_GRADS", "1") != "0"


class Conv2d(nn.Module):
    """nn.Conv2d(bias optional) -> scouter_conv2d_{fwd,dgrad,wgrad}_f32.  Per-group channels must be multiples of 32."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.groups = stride, padding, groups
        k = kernel_size
        w = torch.empty(k, k, in_channels // groups, out_channels)
        self.weight = nn.Parameter(w.permute(3, 2, 0, 1))          # logical OIHW, physical HWIO
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self._dw = self._db = None                                  # arena slices, set by GradArena
        self.precision = "fp32"                                     # per layer; SlotModel.set_precision (no global)
        self.use_side_stream = K.SIDE_STREAM_DEFAULT                # weight gradient on the side stream
        self.planes = 0                                             # 3: bf16x3 plane kernels (SlotModel.set_planes)
        self._wsplit = None                                         # (wf, wd) of this step, from the model's one-launch split
        self.x3 = 0                                                 # bits of kernels.X3_DEFAULT (SlotModel.set_x3): register-split bf16x3 GEMM
        self._capture = None                                        # test instrumentation, see BatchNorm2d
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")   # timm resnet.py:447-448
        if bias:
            bound = 1.0 / math.sqrt(in_channels // groups * k * k)
            nn.init.uniform_(self.bias, -bound, bound)

    # ---- pointwise layers on the register-split bf16x3 GEMM (csrc/conv_x3.hip): fp32 tensors in, weight planes from the
    # model's one split launch
    def x3_static(self):
        """Shape rule alone (kernels.x3_eligible): 1x1 / stride 1 / no bias / deep enough."""
        k = self.kernel_size
        return K.x3_eligible(self.in_channels, self.out_channels, k, k, self.stride, self.padding, self.groups,
                             self.bias is not None)

    def x3_conv_static(self):
        """Shape rule alone (kernels.x3_conv_eligible): 3x3 / stride 1 / pad 1 with 32 channels per group."""
        k = self.kernel_size
        return K.x3_conv_eligible(self.in_channels, self.out_channels, k, k, self.stride, self.padding, self.groups,
                                  self.bias is not None)

    def xpw_static(self):
        """Shape rule alone: the FORWARD of this pointwise layer runs on the persistent bf16x3 kernel (kernels.xpw_fwd_eligible;
        csrc/conv_pw_persist_x3.h) -- 64 / 128 input channels always, 256 where the register-split GEMM does not serve the
        layer (it wins from 512 output channels on: tools_dev/xpw_fwd_bench.py)."""
        k = self.kernel_size
        return (K.xpw_fwd_eligible(self.in_channels, self.out_channels, k, k, self.stride, self.padding, self.groups,
                                   self.bias is not None) and (self.in_channels <= 128 or not self.x3_static()))

    def x3_halo_static(self):
        """(forward, input gradient) on the persistent resident-rows kernel -- shape rule alone (kernels.x3_halo_eligible)."""
        k = self.kernel_size
        return K.x3_halo_eligible(self.in_channels, self.out_channels, k, k, self.stride, self.padding, self.groups,
                                  self.bias is not None)

    def halo_fwd(self):
        """x3 bit 7 (opt-in), fp32 precision, no plane operands: the FORWARD of this 3x3 layer has 32 output channels per group
        and runs on csrc/conv_xhalo.hip (tile 7 of the register-split entry points)."""
        return bool(self.x3 & 128) and self.precision == "fp32" and not self.planes and self.x3_halo_static()[0]

    def halo_dgrad(self):
        """x3 bit 6, fp32 precision: the INPUT GRADIENT of this 3x3 layer has 32 input channels per group and runs on
        csrc/conv_xhalo.hip from the fp32 output gradient (plane layers included: their gradient planes need 64)."""
        return bool(self.x3 & 64) and self.precision == "fp32" and self.x3_halo_static()[1]

    def fwd_on_xpw(self):
        """x3 bit 5, fp32 precision, no plane operands, the shape rule."""
        return bool(self.x3 & 32) and self.precision == "fp32" and not self.planes and self.xpw_static()

    def x3_mode(self):
        """Bits of `x3` this layer uses now (fp32 precision only): bits 0-3 for a pointwise layer under the static shape
        rule (and no plane operands), 16 for a 3x3 layer with 32-channel groups and no plane operands (forward only:
        kernels.x3_conv_eligible); 0 otherwise."""
        if not self.x3 or self.precision != "fp32":
            return 0
        if not self.planes and self.x3_static():
            return self.x3 & 15
        bits = 0
        if (self.x3 & 16) and not self.planes and self.x3_conv_static():
            bits |= 16
        if not self.planes and (self.halo_fwd() or self.halo_dgrad()):
            bits |= 64                    # (3x3 layers with 32 GEMM columns per group: csrc/conv_xhalo.hip; forward x3 bit 7,
            #                               input gradient x3 bit 6)
        return bits

    def _x3_weights(self, want_fwd, want_dgrad):
        ws, self._wsplit = self._wsplit, None
        if ws is not None and (ws[0] is not None or not want_fwd) and (ws[1] is not None or not want_dgrad):
            return ws
        return K.planes_split_weight(K.hwio(self.weight), self.groups, 3, fwd=want_fwd, dgrad=want_dgrad)

    # ---- bf16x3 operand planes (csrc/conv_planes.hip): the producer of this layer's input hands over a K.PlaneTensor
    def _nplanes(self):
        """3 exact bf16 planes per operand in fp32 mode (fp32-grade products), 1 (the RNE-rounded value: what the
        bf16-input kernels compute, without their in-flight conversion and with half the operand bytes) in bf16 mode."""
        return 0 if not self.planes else (self.planes if self.precision == "fp32" else 1)

    def planes_in(self):
        """Number of operand planes this layer wants its INPUT in (0: plain fp32 tensor)."""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        n = self._nplanes()
        return n if n and self.stride == 1 and cg % 32 == 0 and ng % 64 == 0 else 0

    def planes_dy(self):
        """Number of planes this layer wants its OUTPUT GRADIENT in (input-gradient kernel on planes)."""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        n = self._nplanes()
        return n if n and self.stride == 1 and ng % 32 == 0 and cg % 64 == 0 else 0

    def planes_only(self, H, W, B=None):
        """True if forward, input gradient AND weight gradient of this layer all run on planes for an H x W map: its
        producers then need not write the fp32 copy of the tensor at all.  In bf16 mode (ONE plane = the bf16-stored
        tensor) that also holds when the gradients run on the bf16-input igemm kernels, which read that plane as a
        bf16-stored activation (32-channel groups: the first grouped layer; needs the batch size for their pixel rule)."""
        if not (self.bias is None and self.planes_in()):
            return False
        if self.planes_dy() and self.planes_wgrad() and 32 // W + 1 < H:
            return True
        return bool(B is not None and self._nplanes() == 1 and self._igemm_bf16_grads(B, H, W))

    def _igemm_bf16_grads(self, B, H, W):
        """input- and weight-gradient both on the bf16-input igemm kernels for a B x H x W map (kernels.conv2d_dgrad /
        conv2d_wgrad rules): they read bf16-stored operands and round fp32 ones to the same values"""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        k = self.kernel_size
        return bool(GRAD_STORAGE_BF16 and self.precision == "bf16" and self.bias is None and self.stride == 1 and
                    2 * self.padding == k - 1 and B * H * W >= K.BF16_MIN_PIXELS and cg % 32 == 0 and ng % 32 == 0 and
                    (k == 1 or 64 // W + 1 < H))

    def dy_plane_only(self, B, H, W):
        """bf16 mode, a plane layer whose input gradient does NOT run on planes (32-channel groups): its output gradient
        can still be handed over as ONE bf16 plane only -- the igemm kernels read it as a bf16-stored tensor."""
        return bool(self.planes and self._nplanes() == 1 and not self.planes_dy() and self._igemm_bf16_grads(B, H, W))

    def planes_wgrad(self):
        """Weight gradient on planes too (same-size convolution, 64-multiples of channels per group)."""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        return bool(self.planes_in() and self.planes_dy() and cg % 64 == 0 and ng % 64 == 0
                    and 2 * self.padding == self.kernel_size - 1)

    def grad_storage(self, B, H, W):
        """Storage type for this layer's OUTPUT gradient (B x H x W output pixels): bf16 when both of its readers -- the
        input-gradient and the weight-gradient kernel -- are the bf16-input ones, which round an fp32 gradient to exactly
        the stored values (so this storage changes no result); None (fp32) otherwise."""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        k = self.kernel_size
        ok = (GRAD_STORAGE_BF16 and self.precision == "bf16" and not self.planes and self.bias is None and
              self.stride == 1 and 2 * self.padding == k - 1 and B * H * W >= K.BF16_MIN_PIXELS and
              cg % 32 == 0 and ng % 32 == 0 and (k == 1 or 64 // W + 1 < H))
        return K.BF16 if ok else None

    def act_storage(self, B, H, W):
        """Storage type for this layer's INPUT activation (B x H x W pixels) when nothing else reads it: bf16 when the
        forward and the weight-gradient kernel are the bf16-input ones (which round an fp32 input to exactly the stored
        values: no result changes); None (fp32) otherwise."""
        cg, ng = self.in_channels // self.groups, self.out_channels // self.groups
        k = self.kernel_size
        ok = (GRAD_STORAGE_BF16 and self.precision == "bf16" and not self.planes and self.stride == 1 and
              2 * self.padding == k - 1 and B * H * W >= K.BF16_MIN_PIXELS and cg % 32 == 0 and ng % 32 == 0 and
              (k == 1 or 64 // W + 1 < H))
        return K.BF16 if ok else None

    def fwd(self, x, save, relu=False, addend=None, bn_stats=False, out_dtype=None):
        """bn_stats=True (a train-mode BatchNorm follows): returns ((y, stats), ctx) -- see kernels.conv2d_fwd.
        out_dtype=torch.bfloat16: the output is STORED as bf16 (activation storage of the bf16 mode, kernels.conv2d_fwd)."""
        if isinstance(x, K.PlaneTensor):
            k = self.kernel_size
            want_wd = save and (self.planes_dy() > 0 or self.halo_dgrad())
            ws, self._wsplit = self._wsplit, None
            if ws is not None and ws[0] is not None and ws[0].shape[0] == x.planes.shape[0] and (ws[1] is not None or not want_wd):
                wf, wd = ws[0], (ws[1] if want_wd else None)        # split by SlotModel for the whole model in one launch
            else:
                wf, wd = K.planes_split_weight(K.hwio(self.weight), self.groups, x.planes.shape[0], fwd=True, dgrad=want_wd)
            y = K.conv2d_fwd_planes(x.planes, wf, k, k, self.stride, self.padding, self.groups, self.bias, addend, relu,
                                    bn_stats, out_dtype=out_dtype or K.F32)
            if self._capture is not None and relu:
                self._capture[0][self._capture[1]] = y
            return y, ((x.f32 if x.f32 is not None else tuple(x.shape), wd,
                        x.planes if self.planes_wgrad() or x.f32 is None else None) if save else None)
        xm = self.x3_mode()
        # (the persistent bf16x3 forward has the plain epilogue only; a call with an addend / ReLU takes the other kernels)
        xpw = 5 if (self.fwd_on_xpw() and addend is None and not relu and x.dtype == K.F32 and out_dtype in (None, K.F32)) else None
        if (xm & 87) and x.dtype == K.F32 and out_dtype in (None, K.F32):
            hf = bool(xm & 64) and self.halo_fwd() and addend is None and x.shape[2] <= K.X3_HALO_MAX_W
            want_wd = bool(save and ((xm & 6) or ((xm & 64) and self.halo_dgrad())))
            want_wf = (bool(xm & 17) and xpw is None) or hf
            wf, wd = self._x3_weights(want_wf, want_wd) if (want_wf or want_wd) else (None, None)
            if xpw is not None:
                y = K.conv2d_fwd(x, K.hwio(self.weight), None, None, 1, 0, 1, False, bn_stats, tile=5)
            elif hf:
                y = K.conv2d_fwd_x3(x, wf, None, relu, bn_stats, tile=7, kh=3, pad=1, groups=self.groups)
            elif xm & 17:
                k = self.kernel_size
                y = K.conv2d_fwd_x3(x, wf, addend, relu, bn_stats, kh=k, pad=self.padding, groups=self.groups)
            else:
                y = K.conv2d_fwd(x, K.hwio(self.weight), self.bias, addend, self.stride, self.padding, self.groups, relu,
                                 bn_stats, precision=self.precision)
            if self._capture is not None and relu:
                self._capture[0][self._capture[1]] = y
            return y, ((x, wd, None) if save else None)
        y = K.conv2d_fwd(x, K.hwio(self.weight), self.bias, addend, self.stride, self.padding, self.groups, relu,
                         bn_stats, precision=self.precision, out_dtype=out_dtype or K.F32, tile=xpw)
        if self._capture is not None and relu:
            self._capture[0][self._capture[1]] = y
        return y, (x if save else None)

    def bwd(self, dy, ctx, need_dx=True, addend=None, post=None, dx_dtype=None):
        """post (K.BnBwdFuse): the input gradient is the gradient of a BatchNorm(+ReLU) output -- its producer finishes
        that BatchNorm's backward reductions in the epilogue (only when dx is computed at all).  dx_dtype=torch.bfloat16:
        that masked gradient is stored as bf16 when the bf16-input kernel fuses (kernels.conv2d_dgrad; check dx.dtype)."""
        wd = xp = None
        if isinstance(ctx, tuple):
            x, wd, xp = ctx
        else:
            x = ctx
        dyp = None
        if isinstance(dy, K.PlaneTensor):
            dy, dyp = dy.f32, dy.planes
        xshape = x if isinstance(x, tuple) else tuple(x.shape)       # (a shape only: the producer wrote planes only)
        xt = None if isinstance(x, tuple) else x
        if xt is None and xp is not None and xp.shape[0] == 1 and not self.planes_wgrad():
            xt = xp[0]                    # (bf16 mode: the one plane IS the bf16-stored activation of the igemm kernels)
        if dy is None and dyp is not None and dyp.shape[0] == 1 and wd is None:
            dy, dyp = dyp[0], None        # (likewise the output gradient, Conv2d.dy_plane_only)
        dev = dy.device if dy is not None else dyp.device
        if need_dx and dyp is not None and wd is not None and self.planes_dy():
            k = self.kernel_size
            dx = K.conv2d_dgrad_planes(dyp, wd, xshape, k, k, self.stride, self.padding, self.groups, addend, post=post)
            need_dx = False
        else:
            dx = None
        if self._dw is not None or self._db is not None:
            # weight / bias gradients: off the critical path
            with K.side_stream(dev, xt, dy, xp, dyp, enabled=self.use_side_stream):
                # (the plane kernel's branch-free pixel walk needs maps that are not tiny: 32 // W + 1 < H)
                if (self._dw is not None and xp is not None and dyp is not None and self.planes_wgrad() and
                        32 // xshape[2] + 1 < xshape[1]):
                    K.conv2d_wgrad_planes(xp, dyp, self._dw, self.padding, self.groups)
                elif (self._dw is not None and (self.x3_mode() & 8) and xt is not None and dy is not None and
                      xt.dtype == K.F32 and dy.dtype == K.F32 and K.x3_wgrad_ok(self.in_channels, self.out_channels)):
                    K.conv2d_wgrad_x3(xt, dy, self._dw)      # (pointwise layer: register-split bf16x3 GEMM, csrc/conv_x3.hip)
                elif self._dw is not None:
                    if xt is None or dy is None:
                        raise RuntimeError("Conv2d.bwd: the fp32 operands were dropped (planes_only) but the weight "
                                           "gradient cannot run on planes for this shape")
                    K.conv2d_wgrad(xt, dy, self._dw, self.stride, self.padding, self.groups, precision=self.precision)
                if self._db is not None:
                    K.colsum(dy, self._db)
        if not need_dx:
            return dx
        if (self.halo_dgrad() and wd is not None and dy is not None and dy.dtype == K.F32 and xshape[2] <= K.X3_HALO_MAX_W and
                dx_dtype in (None, K.F32)):
            # 3x3 layer with 32 input channels per group: resident rows, split once per element (csrc/conv_xhalo.hip)
            return K.conv2d_dgrad_x3(dy, wd, xshape, addend, post=post, tile=7, kh=3, pad=1, groups=self.groups)
        if self.x3_mode() and wd is not None and dyp is None and dy is not None and dy.dtype == K.F32:
            # pointwise layer on the register-split bf16x3 GEMM (x3_mode): plain input gradient (bit 1), or with the
            # fused BatchNorm-backward epilogue (bit 2)
            xm, k = self.x3_mode(), self.kernel_size
            fused = post is not None and K._fuse_wanted(post, k) and post.x_io() == 0
            if ((xm & 4) and self.out_channels >= K.X3_FUSED_MIN_K) if fused else (xm & 2):
                return K.conv2d_dgrad_x3(dy, wd, xshape, addend, post=post, kh=k, pad=self.padding, groups=self.groups)
        return K.conv2d_dgrad(dy, K.hwio(self.weight), xshape, addend, self.stride, self.padding, self.groups,
                              precision=self.precision, post=post, out_dtype=dx_dtype or K.F32)


class StemConv2d(Conv2d):
    """First convolution of the network (Cin = 1 or 3, NCHW image in): im2col into [M][Kpad] rows (zero padded to a
    multiple of 32) + the generic MFMA GEMM, for the forward and the weight gradient.  No input gradient.
    SCOUTER_STEM_DIRECT=1 (opt-in, kernels.STEM_DIRECT): the deep stem's 3 -> 32 layer runs its forward as one direct pass."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        nn.Module.__init__(self)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.groups = stride, padding, 1
        k = kernel_size
        self.kdim = k * k * in_channels
        self.kpad = (self.kdim + 31) // 32 * 32
        self.weight = nn.Parameter(torch.empty(k, k, in_channels, out_channels).permute(3, 2, 0, 1))
        self.bias = None
        self._dw = self._db = None
        self.precision = "fp32"
        self.use_side_stream = K.SIDE_STREAM_DEFAULT
        self.planes = 0
        self.x3 = 0
        self._saved_image = False
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def fwd(self, x_nchw, save, relu=False, addend=None, bn_stats=False):
        if x_nchw.dim() != 4 or x_nchw.shape[1] != self.in_channels:
            raise RuntimeError("expected an NCHW image batch with %d channel(s), got shape %s"
                               % (self.in_channels, tuple(x_nchw.shape)))
        if (K.STEM_DIRECT and self.precision == "fp32" and not relu and addend is None and x_nchw.dtype == torch.float32 and
                K.stem_direct_eligible(self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
                                       x_nchw.shape[3])):
            # deep stem (3 -> 32, 3x3 / 2), SCOUTER_STEM_DIRECT=1: one direct pass over the image, no patch rows; the weight
            # gradient builds its patch rows itself, on its side stream (bwd)
            y = K.stem_direct_fwd(x_nchw, K.hwio(self.weight), bn_stats)
            self._saved_image = True
            return y, (x_nchw if save else None)
        self._saved_image = False
        col = K.im2col_nchw(x_nchw, self.kernel_size, self.stride, self.padding, self.kpad)
        wflat = K.hwio(self.weight).reshape(-1)
        wpad = K.pad_rows(wflat, wflat.numel(), self.kpad * self.out_channels).view(1, 1, self.kpad, self.out_channels)
        y = K.conv2d_fwd(col, wpad, None, addend, 1, 0, 1, relu, bn_stats, precision=self.precision)
        return y, (col if save else None)

    def bwd(self, dy, ctx, need_dx=False, addend=None):
        if need_dx:
            raise NotImplementedError("gradient w.r.t. the input image is not part of the training hot path")
        if self._dw is not None:
            with K.side_stream(dy.device, ctx, dy, enabled=self.use_side_stream):
                if self._saved_image:             # (the direct forward saved the image, not patch rows)
                    ctx = K.im2col_nchw(ctx, self.kernel_size, self.stride, self.padding, self.kpad)
                dwpad = torch.empty((1, 1, self.kpad, self.out_channels), dtype=torch.float32, device=dy.device)
                K.conv2d_wgrad(ctx, dy, dwpad, precision=self.precision)
                n = self.kdim * self.out_channels
                K.axpby(dwpad.view(-1)[:n], None, 1.0, 0.0, out=self._dw.reshape(-1))
        return None


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d (eps 1e-5, momentum 0.1) with ReLU / residual-add fused into the apply pass."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._dg = self._db = None
        self._capture = None                 # test instrumentation: (dict, key) -> the ReLU'd output is stored there

    def fwd(self, x, save, relu=False, residual=None, tracked=None, planes=0, residual_bn=None, keep_f32=True,
            out_dtype=None):
        """x may be the (tensor, stats) pair a conv produced with bn_stats=True.  out_dtype=torch.bfloat16: the output is
        stored as bf16 (x / residual may be bf16-stored too: kernels.bn_fwd).  planes = 1 / 3: the output is a
        K.PlaneTensor (fp32 + bf16 operand planes for the plane convolution that consumes it).  residual_bn: `residual`
        is the raw output of the downsample convolution and this the saved block of ITS BatchNorm (stats_only): both
        BatchNorms are applied in this one pass."""
        stats = None
        if isinstance(x, tuple):
            x, stats = x
        if self.training and x.numel() == x.shape[-1]:     # torch.nn.functional.batch_norm's check (batch 1 on a 1x1 map)
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(x.shape),))
        out = K.bn_fwd(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, relu,
                       residual, self.momentum, self.eps, stats if self.training else None,
                       want_mask=bool(relu and save), planes=planes, residual_bn=residual_bn, keep_f32=keep_f32,
                       out_dtype=out_dtype or K.F32)
        if self.training and tracked is not None:
            tracked.append(self.num_batches_tracked)
        if self._capture is not None and relu:
            o = out[0].f32 if isinstance(out[0], K.PlaneTensor) else out[0]
            self._capture[0][self._capture[1]] = o if o is not None else K.bn_apply(x, out[1], True)
        # the backward takes the ReLU sign from a 1-bit/element mask, not from the 4-byte activation
        return out[0], ((x, out[2] if relu else None, out[1], self.training) if save else None)

    def stats_only(self, x, tracked=None, relu_follows=True):
        """Statistics / running-stat update without the apply pass -> (raw x, saved [4, C]); the consumer evaluates
        relu((x - mean) * scale + shift) itself (fused split attention, timm/models/layers/split_attn.py)."""
        stats = None
        if isinstance(x, tuple):
            x, stats = x
        if self.training and x.numel() == x.shape[-1]:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(x.shape),))
        saved = K.bn_stats(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                           self.momentum, self.eps, stats if self.training else None)
        if self.training and tracked is not None:
            tracked.append(self.num_batches_tracked)
        if self._capture is not None and relu_follows:   # test instrumentation: the activation the fused kernels see
            self._capture[0][self._capture[1]] = K.bn_apply(x, saved, True)
        return x, saved

    @staticmethod
    def fuse(ctx, *more):
        """K.BnBwdFuse for the kernel that produces this BatchNorm's output gradient (`more`: contexts of further
        BatchNorms fed by the same gradient -- the downsample branch)."""
        return K.BnBwdFuse(ctx[1], [(c[0], c[2]) for c in (ctx,) + more])

    def bwd(self, dy, ctx, want_gout=False, fused=None, dx_dtype=None):
        """fused = BnBwdFuse.ext(i): dy is already masked and its sums are reduced (by the producer's epilogue).
        dx_dtype=torch.bfloat16: dx is stored as bf16 (Conv2d.grad_storage of the convolution in front)."""
        x, mask, saved, training = ctx
        if fused is not None:
            return K.bn_bwd(dy, None, x, saved, training, self._dg, self._db, want_gout, ext=fused,
                            dx_dtype=dx_dtype or K.F32)
        return K.bn_bwd(dy, None, x, saved, training, self._dg, self._db, want_gout, mask=mask, dx_dtype=dx_dtype or K.F32)


class LinearParams(nn.Module):
    """Parameter holder with nn.Linear's names / init; the arithmetic runs in the HIP kernels of the owner."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))
        self._g = {}

    def _bind_grad(self, leaf, seg):
        self._g[leaf] = seg.view(self.weight.shape) if leaf == "weight" else seg


class GradArena:
    """One flat fp32 buffer holding the gradient of every trainable parameter, in named_parameters() order.
    `param.grad` are views into it (conv weights with the parameter's own HWIO strides).  Parameters that never
    receive a gradient on this path (slot.to_q.*, reference slot_attention.py:52-53) are excluded, like the
    `find_unused_parameters=True` of the reference's DDP wrap (train.py:140)."""

    def __init__(self, model, exclude=("to_q",)):
        self.entries = []       # (name, param, offset, numel)
        off = 0
        for name, p in model.named_parameters():
            if not p.requires_grad or any(e in name for e in exclude):
                continue
            self.entries.append((name, p, off, p.numel()))
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        dev = self.entries[0][1].device if self.entries else torch.device("cpu")
        self.flat = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        self.views = {}
        for name, p, o, n in self.entries:
            seg = self.flat[o:o + n]
            if p.dim() == 4:
                co, cg, kh, kw = p.shape
                self.views[name] = seg.view(kh, kw, cg, co).permute(3, 2, 0, 1)
            elif p.dim() == 2 and getattr(p, "_hip_transposed", False):
                self.views[name] = seg.view(p.shape[1], p.shape[0]).t()
            else:
                self.views[name] = seg.view(p.shape)
        self._bind(model)

    def _bind(self, model):
        mods = dict(model.named_modules())
        for name, p, o, n in self.entries:
            mod_name, _, leaf = name.rpartition(".")
            m = mods[mod_name]
            seg = self.flat[o:o + n]
            if isinstance(m, Conv2d):
                if leaf == "weight":
                    co, cg, kh, kw = p.shape
                    m._dw = seg.view(kh, kw, cg, co)
                else:
                    m._db = seg
            elif isinstance(m, BatchNorm2d):
                if leaf == "weight":
                    m._dg = seg
                else:
                    m._db = seg
            else:
                hook = getattr(m, "_bind_grad", None)
                if hook is None:
                    raise RuntimeError("no gradient binding for parameter %s" % name)
                hook(leaf, seg)

    def first_offset(self, prefix):
        """Arena offset of the first parameter whose name starts with `prefix` (None if there is none)."""
        for name, p, o, n in self.entries:
            if name.startswith(prefix):
                return o
        return None

    def attach(self):
        """Expose the arena slices as param.grad (PyTorch semantics: .grad holds the gradient of the last backward)."""
        for name, p, o, n in self.entries:
            p.grad = self.views[name]

    def matches(self, model):
        cur = [(n, p) for n, p in model.named_parameters() if p.requires_grad and "to_q" not in n]
        return len(cur) == len(self.entries) and all(a[1] is b[1] and b[1].device == self.flat.device
                                                     for a, b in zip(cur, self.entries))
