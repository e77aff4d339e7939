"""csrc/conv_halo_dgrad_bf16.h (-m gpu): the resident-rows input gradient of the 3x3 / stride 1 / pad 1 layers with 32 input
channels per group in bf16 mode (BASELINE configs[4]; reference call sites: the deep stem, timm/models/resnet.py:471-489; layer1's
radix convolution, timm/models/layers/split_attn.py:54-60).  Same operands as the 128 x 32 tile kernel it replaces
(SCOUTER_BHALO=0) -- dy and the weights rounded to bf16 (RNE), fp32 accumulation -- in another summation order: against an fp64
input gradient of the SAME rounded values it is at least as close as that kernel; ragged shapes (maps 3 ... 112 pixels wide, a
last tile that ends inside an image, one ... three groups, 32 / 64 dy channels per group), the addend, storage neutrality (an
fp32-stored dy meets the same kernel through one rounding pass) and the fused BatchNorm-backward epilogue against the tile kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ref64(dy, w_hwio, x_shape, groups):
    """fp64 input gradient of the bf16-rounded operands"""
    B, H, W, Cin = x_shape
    wr = w_hwio.to(BF16).double().cpu().permute(3, 2, 0, 1).contiguous()          # OIHW
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, wr, padding=1, groups=groups)
    (y * dy.to(BF16).double().cpu().permute(0, 3, 1, 2)).sum().backward()
    return x.grad.permute(0, 2, 3, 1).contiguous()


def _dgrad(monkeypatch, flag, dy, w, x_shape, groups, addend=None, post=None):
    from scouter_amd import kernels as K
    monkeypatch.setenv("SCOUTER_BHALO", "1" if flag else "0")
    monkeypatch.setattr(K, "BHALO", bool(flag))
    dx = K.conv2d_dgrad(dy, w, x_shape, addend, 1, 1, groups, precision="bf16", post=post)
    torch.cuda.synchronize()
    return dx


SHAPES = [(2, 24, 24, 64, 128, 2),       # layer1's radix convolution: two groups x 64 dy channels (two halves)
          (3, 20, 19, 32, 32, 1),        # 32 dy channels, a last tile that ends inside an image
          (1, 12, 112, 32, 64, 1),       # the widest rows the image holds
          (9, 14, 9, 96, 96, 3),         # three groups
          (40, 9, 3, 32, 64, 1),         # 3-pixel rows: every pixel on a border
          (2, 112, 112, 32, 64, 1)]      # the deep stem's map


@pytest.mark.parametrize("cfg", SHAPES)
def test_input_gradient_is_as_close_to_fp64_as_the_tile_kernel(cfg, monkeypatch):
    B, H, W, Cin, Cout, groups = cfg
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg))
    dy = (torch.randn(B, H, W, Cout, device="cuda", generator=gen)).to(BF16)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=gen) * 0.1
    ref = _ref64(dy, w, (B, H, W, Cin), groups)
    dh = _dgrad(monkeypatch, True, dy, w, (B, H, W, Cin), groups)
    dt = _dgrad(monkeypatch, False, dy, w, (B, H, W, Cin), groups)
    assert dh.dtype == torch.float32 and torch.isfinite(dh).all()
    eh = float((dh.cpu().double() - ref).abs().max())
    et = float((dt.cpu().double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert eh <= max(1.5 * et, 2e-6 * scale), (eh, et, scale)
    assert torch.equal(dh, _dgrad(monkeypatch, True, dy, w, (B, H, W, Cin), groups))


def test_addend_and_storage_neutrality(monkeypatch):
    B, H, W, Cin, Cout, groups = 3, 20, 19, 64, 128, 2
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    dy32 = torch.randn(B, H, W, Cout, device="cuda", generator=gen)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=gen) * 0.1
    add = torch.randn(B, H, W, Cin, device="cuda", generator=gen)
    a = _dgrad(monkeypatch, True, dy32.to(BF16), w, (B, H, W, Cin), groups, addend=add)
    b = _dgrad(monkeypatch, True, dy32, w, (B, H, W, Cin), groups, addend=add)          # fp32-stored dy: rounded, same kernel
    c = _dgrad(monkeypatch, True, dy32.to(BF16).float(), w, (B, H, W, Cin), groups, addend=add)
    assert torch.equal(a, b) and torch.equal(a, c)
    plain = _dgrad(monkeypatch, True, dy32.to(BF16), w, (B, H, W, Cin), groups)
    assert float((a - (plain + add)).abs().max()) <= 1e-5 * float(a.abs().max())
    t = _dgrad(monkeypatch, False, dy32.to(BF16), w, (B, H, W, Cin), groups, addend=add)
    assert float((a - t).abs().max()) <= 2e-6 * float(t.abs().max()) * np.sqrt(9 * 64 / 16.0)


@pytest.mark.parametrize("x_bf16", [False, True])
def test_fused_batchnorm_backward_epilogue_matches_the_tile_kernel(x_bf16, monkeypatch):
    """The epilogue is shared (igemm_epilogue_typed): masked gradient and partial sums agree with the tile kernel's to the
    rounding of the different K order."""
    from scouter_amd import kernels as K
    B, H, W, Cin, Cout, groups = 4, 24, 24, 64, 128, 2
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=gen).to(BF16)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=gen) * 0.1
    x1 = torch.randn(B, H, W, Cin, device="cuda", generator=gen)
    gamma, beta = torch.rand(Cin, device="cuda", generator=gen) + 0.5, torch.randn(Cin, device="cuda", generator=gen)
    rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
    _, saved, mask = K.bn_fwd(x1, gamma, beta, rm, rv, True, True, want_mask=True)
    xs = x1.to(BF16) if x_bf16 else x1
    res = []
    for flag in (True, False):
        post = K.BnBwdFuse(mask, [(xs, saved)])
        g = _dgrad(monkeypatch, flag, dy, w, (B, H, W, Cin), groups, post=post)
        assert post.applied
        res.append((g, post.parts[0].sum(0)))
    (gh, ph), (gt, pt) = res
    sc = float(gt.abs().max())
    assert float((gh - gt).abs().max()) <= 4e-6 * sc
    assert bool(((gh == 0) == (gt == 0)).all())                      # the same ReLU mask
    np.testing.assert_allclose(ph.cpu().numpy(), pt.cpu().numpy(), rtol=1e-5, atol=1e-4 * sc)


FWD_SHAPES = [(2, 24, 24, 64, 128, 2), (3, 20, 19, 32, 32, 1), (1, 12, 112, 32, 64, 1), (9, 14, 9, 96, 96, 3), (40, 9, 3, 32, 64, 1),
              (2, 112, 112, 32, 64, 1)]


@pytest.mark.parametrize("cfg", FWD_SHAPES)
def test_forward_is_bit_identical_to_the_tile_kernel(cfg, monkeypatch):
    """bhalo_fwd_kernel keeps the tile kernel's MFMA sequence per accumulator (tap outer, k-step inner): output, fused BatchNorm
    statistics rows and the bf16-stored output agree bit for bit; an fp32-stored x (the tile kernel) gives the same bits too."""
    from scouter_amd import kernels as K
    B, H, W, Cin, Cout, groups = cfg
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg) + 1)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=gen).to(BF16)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=gen) * 0.1
    res = {}
    for flag in (True, False):
        monkeypatch.setenv("SCOUTER_BHALO", "1" if flag else "0")
        monkeypatch.setattr(K, "BHALO", flag)
        if not flag:                     # the same 128-pixel tile, so that the statistics rows line up
            K._tile_cache[("fwd", True, B, H, W, Cin, Cout, 3, 3, 1, 1, groups)] = 1 if Cout // groups == 64 else 3
        y, (p, rows) = K.conv2d_fwd(x, w, None, None, 1, 1, groups, False, bn_stats=True, precision="bf16")
        yb = K.conv2d_fwd(x, w, None, None, 1, 1, groups, False, precision="bf16", out_dtype=BF16)
        yr = K.conv2d_fwd(x, w, None, None, 1, 1, groups, True, precision="bf16")
        torch.cuda.synchronize()
        res[flag] = (y, p[:rows].clone(), yb, yr)
    K._tile_cache.clear()
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][2], res[True][0].to(BF16)) and torch.equal(res[True][3], res[True][0].clamp_min(0))
    monkeypatch.setenv("SCOUTER_BHALO", "1"); monkeypatch.setattr(K, "BHALO", True)
    y32, (p32, rows) = K.conv2d_fwd(x.float(), w, None, None, 1, 1, groups, False, bn_stats=True, precision="bf16")
    assert torch.equal(y32, res[True][0]) and torch.equal(p32[:rows], res[True][1])
    ref = torch.nn.functional.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.to(BF16).double().cpu().permute(3, 2, 0, 1),
                                     padding=1, groups=groups).permute(0, 2, 3, 1)
    assert float((res[True][0].cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_library_profile_names_the_kernel(monkeypatch):
    import ctypes
    from scouter_amd import _native
    L = _native.lib()
    buf = ctypes.create_string_buffer(1 << 14)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    dy = torch.randn(8, 16, 16, 64, device="cuda", generator=gen).to(BF16)
    w = torch.randn(3, 3, 32, 64, device="cuda", generator=gen)
    from scouter_amd import kernels as K
    for flag, name, fname in ((True, "bhalo_dgrad<bf16>", "bhalo_fwd<bf16>"), (False, "igemm_dgrad_bf16<128x32>", "igemm_fwd_bf16<")):
        L.scouter_prof_collect(buf, len(buf)); L.scouter_prof_enable(1)
        _dgrad(monkeypatch, flag, dy, w, (8, 16, 16, 32), 1)
        K.conv2d_fwd(dy[..., :32].contiguous(), w, None, None, 1, 1, 1, precision="bf16")
        torch.cuda.synchronize()
        L.scouter_prof_enable(0); L.scouter_prof_collect(buf, len(buf))
        names = [row.split("\t")[0] for row in buf.value.decode().splitlines()]
        assert name in names and any(n.startswith(fname) for n in names), (flag, names)
