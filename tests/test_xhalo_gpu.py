"""csrc/conv_xhalo.hip -- tile 7 of the register-split entry points (-m gpu): the 3x3 passes whose GEMM is 32 columns wide per
group (reference call sites: the deep stem, timm/models/resnet.py:471-489, and layer1's radix convolution,
timm/models/layers/split_attn.py:54-60) on the persistent resident-rows kernel.  Against an fp64 convolution (at least as
close as the exact-fp32 MFMA kernel), the fp32 kernel's fused BatchNorm statistics / BatchNorm-backward sums and ReLU masking,
over ragged shapes (tiles spanning several images, maps 7 ... 126 wide, one ... three groups, 16-multiples of K), more tiles
than workgroups (several rounds of the persistent tile walk) and the fixtures' batch sizes; bit-reproducible."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def K():
    from scouter_amd import kernels
    return kernels


def _rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, device="cuda", generator=gen) * scale


def _ref_fwd(x, w, groups):
    return torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, 1, 1, 1,
                                      groups).permute(0, 2, 3, 1).contiguous()


def _ref_dgrad(dy, w, groups):
    return torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, 1, 1, 0,
                                                groups).permute(0, 2, 3, 1).contiguous()


FWD = [(3, 9, 7, 32, 32, 1), (2, 20, 19, 64, 64, 2), (1, 5, 126, 32, 32, 1), (5, 33, 40, 96, 96, 3), (37, 14, 14, 32, 32, 1),
       (6, 112, 112, 32, 32, 1), (16, 112, 112, 32, 32, 1), (1, 3, 3, 64, 32, 1)]


@pytest.mark.parametrize("cfg", FWD)
@pytest.mark.parametrize("relu", [False, True])
def test_forward_with_fused_statistics(cfg, relu):
    B, H, W, Cin, Cout, groups = cfg
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg))
    x = _rnd(gen, B, H, W, Cin)
    w = _rnd(gen, 3, 3, Cin // groups, Cout, scale=1.0 / np.sqrt(9 * Cin // groups))
    wf, _ = kk.planes_split_weight(w, groups, 3, fwd=True, dgrad=False)
    ref = _ref_fwd(x, w, groups)
    if relu:            # (statistics are taken before any activation: the two do not combine)
        y = kk.conv2d_fwd_x3(x, wf, None, True, False, tile=7, kh=3, pad=1, groups=groups)
        y32 = kk.conv2d_fwd(x, w, None, None, 1, 1, groups, True, False)
        ref = ref.clamp_min(0)
    else:
        y, (part, rows) = kk.conv2d_fwd_x3(x, wf, None, False, True, tile=7, kh=3, pad=1, groups=groups)
        y32, (p32, r32) = kk.conv2d_fwd(x, w, None, None, 1, 1, groups, False, True)
        assert rows == kk._native.lib().scouter_conv2d_x3_halo_partial_rows(groups) and tuple(part.shape) == (rows, Cout, 2)
        yd = y.double().view(-1, Cout)
        s = part[:rows].sum(0)
        torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-9, atol=1e-7)           # fp64 sums of the kernel's own output
        torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)
        y2, (part2, _) = kk.conv2d_fwd_x3(x, wf, None, False, True, tile=7, kh=3, pad=1, groups=groups)
        assert torch.equal(y, y2) and torch.equal(part[:rows], part2[:rows])            # reproducible bit for bit
    e = float((y.double() - ref).abs().max())
    e32 = float((y32.double() - ref).abs().max())
    assert e <= max(e32, 1e-6) * 1.05, (cfg, e, e32)       # (measured: a third of the exact-fp32 MFMA kernel's error)


DGRAD = [(3, 9, 7, 32, 32, 1), (2, 20, 19, 64, 128, 2), (1, 5, 126, 32, 64, 1), (5, 33, 40, 96, 192, 3), (37, 14, 14, 32, 32, 1),
         (6, 112, 112, 32, 64, 1), (6, 56, 56, 64, 128, 2), (16, 112, 112, 32, 32, 1)]


@pytest.mark.parametrize("cfg", DGRAD)
@pytest.mark.parametrize("mode", ["plain", "addend", "bn", "bn+addend", "bn-nomask", "bn2"])
def test_input_gradient_and_its_fused_batchnorm_backward_epilogue(cfg, mode):
    B, H, W, Cin, Cout, groups = cfg
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg) + len(mode))
    dy = _rnd(gen, B, H, W, Cout)
    w = _rnd(gen, 3, 3, Cin // groups, Cout, scale=1.0 / np.sqrt(9 * Cout // groups))
    xs = (B, H, W, Cin)
    add = _rnd(gen, *xs) if "addend" in mode else None
    _, wd = kk.planes_split_weight(w, groups, 3, fwd=False, dgrad=True)
    ref = _ref_dgrad(dy, w, groups)
    if add is not None:
        ref = ref + add.double()
    fused = mode.startswith("bn")
    if fused:
        g_, b_ = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        xb = [_rnd(gen, *xs) for _ in range(2 if mode == "bn2" else 1)]
        outs = [kk.bn_fwd(t, g_, b_, torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda"), True, True, want_mask=True)
                for t in xb]
        mask = None if mode == "bn-nomask" else outs[0][2]
        mk = lambda: kk.BnBwdFuse(mask, [(t, o[1]) for t, o in zip(xb, outs)])
    else:
        mk = lambda: None
    post = mk()
    dx = kk.conv2d_dgrad_x3(dy, wd, xs, add, post=post, tile=7, kh=3, pad=1, groups=groups)
    post_b = mk()
    dx_b = kk.conv2d_dgrad_x3(dy, wd, xs, add, post=post_b, tile=7, kh=3, pad=1, groups=groups)
    assert torch.equal(dx, dx_b)
    if Cout // groups % 32 == 0:
        post32 = mk()
        d32 = kk.conv2d_dgrad(dy, w, xs, add, 1, 1, groups, post=post32)
    else:
        d32, post32 = None, None
    a = dx.double()
    if fused:
        assert post.applied and post.rows == kk._native.lib().scouter_conv2d_x3_halo_partial_rows(groups)
        for p, q in zip(post.parts, post_b.parts):
            assert torch.equal(p[:post.rows], q[:post.rows])
        if mask is not None:
            # the mask the apply pass wrote = sign of its ReLU'd output
            live = (outs[0][0] > 0)
            assert bool((a[~live] == 0).all()) and 0.2 < float(live.float().mean()) < 0.8
            ref = torch.where(live, ref, torch.zeros_like(ref))
        # the sums: fp64 over the kernel's own (masked) gradient
        gsum = a.view(-1, Cin).sum(0)
        for i, t in enumerate(xb):
            sv = outs[i][1]
            xh = ((t - sv[0]) * sv[1]).double().view(-1, Cin)
            s = post.parts[i][:post.rows].sum(0)
            torch.testing.assert_close(s[:, 0], gsum, rtol=1e-9, atol=1e-6)
            torch.testing.assert_close(s[:, 1], (a.view(-1, Cin) * xh).sum(0), rtol=1e-6, atol=1e-4)
        if post32 is not None:
            assert post32.applied
            for p, q in zip(post.parts, post32.parts):
                sa, sb = p[:post.rows].sum(0), q[:post32.rows].sum(0)
                assert float((sa - sb).abs().max()) <= 1e-4 * float(sb.abs().max())
    e = float((a - ref).abs().max())
    if d32 is not None:
        e32 = float((d32.double() - ref).abs().max())
        assert e <= max(e32, 1e-6) * 1.05, (cfg, mode, e, e32)
    else:
        assert e <= 5e-6 * float(ref.abs().max()), (cfg, mode, e)


def test_tile_7_names_what_it_does_not_cover():
    """tile 7 on a shape it does not serve is an error, never a silent re-route"""
    kk = K()
    L = kk._native.lib()
    x = torch.zeros(2, 8, 8, 64, device="cuda")
    w = torch.zeros(3, 3, 64, 64, device="cuda")
    wf, wd = kk.planes_split_weight(w, 1, 3)
    y = torch.empty(2, 8, 8, 64, device="cuda")
    rc = L.scouter_conv2d_fwd_x3(kk._p(x), kk._p(wf), None, None, kk._p(y), None, 2, 8, 8, 64, 64, 3, 3, 1, 1, 0, 7, None)
    assert rc != 0 and "tile 7" in L.scouter_last_error().decode()
    rc = L.scouter_conv2d_dgrad_x3_bnbwd(kk._p(y), kk._p(wd), None, kk._p(x), 2, 8, 8, 64, 64, 3, 3, 1, 1, 7, *kk._NO_FUSE, None)
    assert rc != 0 and "tile 7" in L.scouter_last_error().decode()


def test_model_routes_the_32_column_passes_and_the_switch_turns_them_off():
    """SlotModel.set_x3 bits 6 / 7: which passes of resnest26d run on the resident-rows kernel is a static function of the layer
    shapes -- input gradients of stem convolutions 2 and 3 and of layer1's two radix convolutions (bit 6, default), the forward of
    stem convolution 2 (bit 7, opt-in).  Bit 6 does not touch the forward (bit-identical log-probabilities); every gradient
    agrees with the bit-off path to fp32 rounding noise (weights upstream of the four layers) or exactly (everything else)."""
    import test_model_gpu as T
    from scouter_amd import _native
    res = {}
    for bits in (63, 127, 255):
        m, P, images, labels = T.build("resnest26d_96")
        m.set_x3(bits)
        convs = [c for c in m.backbone.modules() if hasattr(c, "halo_dgrad")]
        assert sum(c.halo_dgrad() for c in convs) == (4 if bits & 64 else 0)
        assert sum(c.halo_fwd() for c in convs) == (1 if bits & 128 else 0)
        m.train()
        L = _native.lib()
        L.scouter_prof_enable(1)
        out, losses = m(images.cuda(), labels.cuda())
        losses[0].backward()
        torch.cuda.synchronize()
        L.scouter_prof_enable(0)
        import ctypes
        buf = ctypes.create_string_buffer(1 << 16)
        L.scouter_prof_collect(buf, len(buf))
        names = {r.split("\t")[0]: float(r.split("\t")[1]) for r in buf.value.decode().splitlines()}
        assert names.get("xhalo_dgrad+bn_bwd<bf16x3>", 0) == (4 if bits & 64 else 0), names
        assert names.get("xhalo_fwd<bf16x3>", 0) == (1 if bits & 128 else 0), names
        res[bits] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(res[63][0], res[127][0])                                    # input gradients do not enter the forward
    assert float((res[255][0] - res[63][0]).abs().max()) < 1e-3
    up = ("backbone.conv1.", "backbone.bn1.", "backbone.layer1.0.conv1.", "backbone.layer1.0.bn1.", "backbone.layer1.1.conv1.",
          "backbone.layer1.1.bn1.", "backbone.layer1.0.downsample", "backbone.layer1.0.conv2.conv.", "backbone.layer1.1.conv2.conv.")
    same = differ = 0
    for k, g in res[63][1].items():
        h = res[127][1][k]
        if k.endswith("conv2.fc1.bias"):          # (a bias in front of a train-mode BatchNorm: exact gradient 0, noise only)
            continue
        if torch.equal(g, h):
            same += 1
        else:
            differ += 1
            assert k.startswith("backbone.conv1.") or k.startswith("backbone.bn1.") or k.startswith("backbone.layer1."), k
            assert float((g - h).abs().max()) <= 2e-3 * float(g.abs().max()) + 1e-6, (k, float((g - h).abs().max()), float(g.abs().max()))
    assert differ >= 4 and same > differ, (same, differ)
