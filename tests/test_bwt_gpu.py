"""csrc/conv_wgrad_taps_bf16.h (-m gpu): the tap-fused weight gradient of the 3x3 / stride 1 / pad 1 layers with 32 input channels
per group when both operands are STORED as bf16 (BASELINE configs[4]'s --precision bf16; reference call sites: the deep stem,
timm/models/resnet.py:471-489; layer1's radix convolution, timm/models/layers/split_attn.py:54-60).  One bf16 product per (tap,
pixel) with fp32 accumulation -- the arithmetic of the per-tap kernel it replaces (SCOUTER_BWT=0), in another summation order:
against an fp64 weight gradient of the SAME bf16 values it is at least as close as that kernel; ragged shapes -- maps 3 ... 112
pixels wide, pixel ranges that span several images and end inside a 64-pixel chunk, one ... three groups, 32 / 64 / 96 / 128
output channels per group --, more pixel ranges than one (split-K slabs); bit-reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref64(x, dy, groups):
    xc = x.permute(0, 3, 1, 2).double().cpu()
    dyc = dy.permute(0, 3, 1, 2).double().cpu()
    w = torch.zeros(dy.shape[-1], x.shape[-1] // groups, 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xc, w, padding=1, groups=groups)
    (y * dyc).sum().backward()
    return w.grad.permute(2, 3, 1, 0).contiguous()          # HWIO


def _wgrad(monkeypatch, flag, x, dy, groups):
    from scouter_amd import kernels as K
    monkeypatch.setenv("SCOUTER_BWT", "1" if flag else "0")
    monkeypatch.setattr(K, "BWT", bool(flag))
    dw = torch.full((3, 3, x.shape[-1] // groups, dy.shape[-1]), float("nan"), device="cuda")
    K.conv2d_wgrad(x, dy, dw, 1, 1, groups, precision="bf16")
    torch.cuda.synchronize()
    return dw


SHAPES = [(8, 12, 12, 32, 32, 1),        # few pixel ranges
          (7, 9, 20, 32, 64, 1),         # images that end inside a chunk
          (8, 14, 10, 64, 128, 2),       # two groups x 64 columns
          (12, 30, 3, 96, 96, 3),        # three groups, 3-pixel rows (every pixel on a border)
          (5, 2, 112, 32, 32, 1),        # the widest rows the ring holds, two rows per image
          (4, 33, 40, 32, 96, 1),        # 96 columns: three 32-column tiles
          (2, 112, 112, 32, 64, 1),      # the deep stem's map: several pixel ranges (split-K slabs)
          (9, 56, 56, 64, 128, 2),       # layer1's radix convolution
          (3, 30, 17, 32, 128, 1)]       # 128 columns per group: two 64-column tiles


@pytest.mark.parametrize("cfg", SHAPES)
def test_weight_gradient_is_as_close_to_fp64_as_the_per_tap_kernel(cfg, monkeypatch):
    B, H, W, Cin, Cout, groups = cfg
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=gen).to(torch.bfloat16)
    dy = (torch.randn(B, H, W, Cout, device="cuda", generator=gen) * 0.1).to(torch.bfloat16)
    ref = _ref64(x, dy, groups)                                      # of the bf16 values: the products are exact in fp32
    dt = _wgrad(monkeypatch, True, x, dy, groups)
    dp = _wgrad(monkeypatch, False, x, dy, groups)
    assert torch.isfinite(dt).all()
    et = float((dt.cpu().double() - ref).abs().max())
    ep = float((dp.cpu().double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert et <= max(1.5 * ep, 3e-7 * scale * np.sqrt(B * H * W / 64.0)), (et, ep, scale)
    again = _wgrad(monkeypatch, True, x, dy, groups)
    assert torch.equal(dt, again)                                   # deterministic split-K


def test_storage_is_bit_neutral(monkeypatch):
    """The rule is a function of the shape only: an fp32-stored operand is rounded to bf16 (RNE) by one extra pass and meets the
    same kernel -- bit for bit the gradient of the bf16-stored operands (the contract of tests/test_storage_gpu.py)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    x = torch.randn(8, 16, 16, 32, device="cuda", generator=gen)
    dy = torch.randn(8, 16, 16, 64, device="cuda", generator=gen) * 0.1
    xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    ref = _wgrad(monkeypatch, True, xb, dyb, 1)
    for a, b in ((x, dyb), (xb, dy), (x, dy), (xb.float(), dyb.float())):
        assert torch.equal(_wgrad(monkeypatch, True, a, b, 1), ref)


def test_library_profile_names_the_kernel(monkeypatch):
    """The static rule routes the shape to the tap-fused kernel (and SCOUTER_BWT=0 away from it): what ran is read from the
    library's own per-kernel profile."""
    import ctypes
    from scouter_amd import _native
    L = _native.lib()
    buf = ctypes.create_string_buffer(1 << 14)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn(8, 16, 16, 32, device="cuda", generator=gen).to(torch.bfloat16)
    dy = torch.randn(8, 16, 16, 64, device="cuda", generator=gen).to(torch.bfloat16)
    for flag, name in ((True, "bwgrad_taps<bf16>"), (False, "wgrad_bf16")):
        L.scouter_prof_collect(buf, len(buf)); L.scouter_prof_enable(1)
        _wgrad(monkeypatch, flag, x, dy, 1)
        L.scouter_prof_enable(0); L.scouter_prof_collect(buf, len(buf))
        names = [row.split("\t")[0] for row in buf.value.decode().splitlines()]
        assert name in names, (flag, names)
