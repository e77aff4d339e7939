"""csrc/conv_wgrad_taps_x3.h (-m gpu): the tap-fused weight gradient of the 3x3 / stride 1 / pad 1 layers with 32 input channels
per group (reference call sites: the deep stem, timm/models/resnet.py:471-489; layer1's radix convolution,
timm/models/layers/split_attn.py:54-60) on the bf16 matrix cores, operands split exactly (hi + mid + lo) in registers.
Against an fp64 weight gradient: at least as close as the exact-fp32 tap-fused kernel it replaces (SCOUTER_XWT=0); ragged
shapes -- maps 3 ... 112 pixels wide, pixel ranges that span several images and end inside a 64-pixel chunk, one ... three groups,
32 / 64 / 96 / 128 output channels per group --, more pixel ranges than one (split-K slabs); bit-reproducible."""
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
    monkeypatch.setenv("SCOUTER_XWT", "1" if flag else "0")
    monkeypatch.setattr(K, "XWT", bool(flag))
    dw = torch.full((3, 3, x.shape[-1] // groups, dy.shape[-1]), float("nan"), device="cuda")
    K.conv2d_wgrad(x, dy, dw, 1, 1, groups)
    torch.cuda.synchronize()
    return dw


SHAPES = [(2, 12, 12, 32, 32, 1),        # one pixel range
          (3, 9, 20, 32, 64, 1),         # images that end inside a chunk
          (2, 14, 10, 64, 128, 2),       # two groups x 64 columns
          (5, 14, 3, 96, 96, 3),         # three groups, 3-pixel rows (every pixel on a border)
          (1, 2, 112, 32, 32, 1),        # the widest rows the ring holds, two rows per image
          (4, 33, 40, 32, 96, 1),        # 96 columns: three 32-column tiles
          (2, 112, 112, 32, 64, 1),      # the deep stem's map: several pixel ranges (split-K slabs)
          (9, 56, 56, 64, 128, 2),       # layer1's radix convolution
          (3, 30, 17, 32, 128, 1)]       # 128 columns per group: two 64-column tiles


@pytest.mark.parametrize("cfg", SHAPES)
def test_weight_gradient_is_as_close_to_fp64_as_the_fp32_kernel(cfg, monkeypatch):
    B, H, W, Cin, Cout, groups = cfg
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(cfg))
    x = torch.randn(B, H, W, Cin, device="cuda", generator=gen)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=gen) * 0.1
    ref = _ref64(x, dy, groups)
    d3 = _wgrad(monkeypatch, True, x, dy, groups)
    d32 = _wgrad(monkeypatch, False, x, dy, groups)
    assert torch.isfinite(d3).all()
    e3 = float((d3.cpu().double() - ref).abs().max())
    e32 = float((d32.cpu().double() - ref).abs().max())
    scale = float(ref.abs().max())
    # fp32-grade: the six-product split is exact, what is left is fp32 accumulation in a different order
    assert e3 <= max(1.5 * e32, 3e-7 * scale * np.sqrt(B * H * W / 64.0)), (e3, e32, scale)
    again = _wgrad(monkeypatch, True, x, dy, groups)
    assert torch.equal(d3, again)                                   # deterministic split-K


def test_library_profile_names_the_kernel(monkeypatch):
    """The static rule routes the shape to the register-split kernel (and SCOUTER_XWT=0 away from it): what ran is read from the
    library's own per-kernel profile."""
    import ctypes
    from scouter_amd import _native
    L = _native.lib()
    buf = ctypes.create_string_buffer(1 << 14)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x = torch.randn(2, 16, 16, 32, device="cuda", generator=gen)
    dy = torch.randn(2, 16, 16, 64, device="cuda", generator=gen)
    for flag, name in ((True, "xwgrad_taps<bf16x3>"), (False, "wgrad_taps")):
        L.scouter_prof_collect(buf, len(buf)); L.scouter_prof_enable(1)
        _wgrad(monkeypatch, flag, x, dy, 1)
        L.scouter_prof_enable(0); L.scouter_prof_collect(buf, len(buf))
        names = [row.split("\t")[0] for row in buf.value.decode().splitlines()]
        assert name in names, (flag, names)
