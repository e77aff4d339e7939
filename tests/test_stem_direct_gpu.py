"""csrc/misc_ops.hip stem_direct_kernel (-m gpu): the deep stem's first convolution -- 3 -> 32 channels, 3x3 / stride 2 /
pad 1, NCHW image in (reference call site: timm/models/resnet.py deep stem conv1[0], used by resnest26d / resnest50d,
timm/models/resnest.py) -- as one direct pass instead of im2col + GEMM.  Against an fp64 convolution (at least as close as
the im2col + exact-fp32 MFMA route), the fused BatchNorm statistics rows, odd / ragged image sizes (odd heights leave a half
row block, odd widths a right border inside the patch), the bench's sizes; the layer's weight gradient (which now builds its
patch rows itself) bit-identical to the im2col route's; bit-reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def K():
    from scouter_amd import kernels
    return kernels


def _ref(x, w_hwio):
    return torch.nn.functional.conv2d(x.double().cpu(), w_hwio.double().cpu().permute(3, 2, 0, 1), None, 2, 1).permute(0, 2, 3, 1)


SHAPES = [(3, 32, 32), (2, 33, 47), (1, 3, 3), (2, 1, 9), (5, 97, 130), (2, 260, 260), (70, 224, 224), (1, 6, 1000)]


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_statistics_rows(shape):
    B, H, W = shape
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(B * 1000 + H * 7 + W)
    x = torch.randn(B, 3, H, W, device="cuda", generator=gen)
    w = torch.randn(3, 3, 3, 32, device="cuda", generator=gen) / np.sqrt(27.0)
    assert kk.stem_direct_eligible(3, 32, 3, 2, 1, W)
    y, (part, rows) = kk.stem_direct_fwd(x, w, bn_stats=True)
    y_plain = kk.stem_direct_fwd(x, w, bn_stats=False)
    assert torch.equal(y, y_plain)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert tuple(y.shape) == (B, Ho, Wo, 32)
    assert rows == kk._native.lib().scouter_stem_direct_partial_rows(B, H) and tuple(part.shape) == (rows, 32, 2)
    ref = _ref(x, w)
    err = float((y.double().cpu() - ref).abs().max())
    # the route it replaces: patch rows + the exact-fp32 MFMA GEMM (three nine-term fmaf chains here, the MFMA's pair chain there)
    col = kk.im2col_nchw(x, 3, 2, 1, 32)
    wpad = kk.pad_rows(w.reshape(-1), w.numel(), 32 * 32).view(1, 1, 32, 32)
    y_gemm = kk.conv2d_fwd(col, wpad, None, None, 1, 0, 1, False, False)
    err_gemm = float((y_gemm.double().cpu() - ref).abs().max())
    assert err <= max(1.5 * err_gemm, 2e-6), (err, err_gemm)
    assert err <= 4e-6 * max(1.0, float(ref.abs().max()))
    yd = y.double().view(-1, 32)
    s = part.sum(0)
    torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-9, atol=1e-7)              # fp64 sums of the kernel's own output
    torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)
    y2, (part2, _) = kk.stem_direct_fwd(x, w, bn_stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)


def test_unsupported_shapes_are_refused_loudly():
    kk = K()
    x = torch.randn(1, 3, 8, 8, device="cuda")
    w = torch.randn(3, 3, 3, 64, device="cuda")
    assert not kk.stem_direct_eligible(3, 64, 3, 2, 1, 8) and not kk.stem_direct_eligible(3, 32, 7, 2, 3, 8)
    assert not kk.stem_direct_eligible(1, 32, 3, 2, 1, 8) and not kk.stem_direct_eligible(3, 32, 3, 1, 1, 8)
    with pytest.raises(RuntimeError):
        kk.stem_direct_fwd(x, w)


@pytest.mark.parametrize("shape", [(4, 64, 64), (3, 45, 51)])
def test_layer_forward_backward_against_the_im2col_route(shape, monkeypatch):
    """nn_hip.StemConv2d: the direct forward saves the IMAGE; the weight gradient makes its patch rows on its own stream --
    the same rows, the same kernel, the same bits as the im2col route."""
    from scouter_amd import nn_hip
    B, H, W = shape
    kk = K()
    torch.manual_seed(11)
    m = nn_hip.StemConv2d(3, 32, 3, 2, 1).cuda()
    x = torch.randn(B, 3, H, W, device="cuda")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Ho, Wo, 32, device="cuda")
    out = {}
    for direct in (True, False):
        monkeypatch.setattr(kk, "STEM_DIRECT", direct)
        for side in (True, False):
            m.use_side_stream = side
            m._dw = torch.zeros(3, 3, 3, 32, device="cuda")
            (y, stats), ctx = m.fwd(x, True, bn_stats=True)
            assert m._saved_image is direct and (ctx is x) is direct
            m.bwd(dy, ctx)
            kk.join_side_stream(x.device)
            torch.cuda.synchronize()
            out[(direct, side)] = (y.clone(), stats[0][:stats[1]].sum(0), m._dw.clone())
    for side in (True, False):
        yd, sd, dwd = out[(True, side)]
        yg, sg, dwg = out[(False, side)]
        assert torch.equal(dwd, dwg)                                                      # same patch rows, same wgrad kernel
        torch.testing.assert_close(yd, yg, rtol=0, atol=4e-6)
        torch.testing.assert_close(sd, sg, rtol=1e-6, atol=1e-4)
    assert torch.equal(out[(True, True)][2], out[(True, False)][2])
    wd = m.weight.detach().double().requires_grad_()
    g, = torch.autograd.grad(torch.nn.functional.conv2d(x.double(), wd, None, 2, 1), wd, dy.double().permute(0, 3, 1, 2))
    got = out[(True, True)][2].double().permute(3, 2, 0, 1)                               # HWIO -> OIHW
    assert float((got - g).abs().max()) <= 2e-5 * max(1.0, float(g.abs().max()))


def test_whole_model_step_with_the_direct_stem(monkeypatch):
    """The smoke step (resnest26d + xSlot, forward + backward against the fp64 oracle, `|HIP - fp64| <= 1.5 x |PyTorch-fp32 - fp64|`)
    with the option on: the direct forward feeds the model, the weight gradient makes its own patch rows."""
    import __graft_entry__ as G
    kk = K()
    monkeypatch.setattr(kk, "STEM_DIRECT", True)
    calls = []
    real = kk.stem_direct_fwd
    monkeypatch.setattr(kk, "stem_direct_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    G.smoke()
    assert calls, "the option did not reach the stem"
