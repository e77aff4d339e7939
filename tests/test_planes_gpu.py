"""Convolution on pre-split bf16 operand planes (scouter_amd/csrc/conv_planes.hip): the three-plane split is exact,
and the six-product bf16 MFMA convolution matches an fp64 convolution as closely as the exact-fp32 MFMA kernel does
(forward incl. padding rows / ragged last tile / groups / fused BatchNorm statistics, and the stride-1 input gradient);
one plane = plain bf16 inputs (BASELINE configs[4])."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def K():
    from scouter_amd import kernels
    return kernels


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().float().cuda()


def test_three_plane_split_is_exact():
    kk = K()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.exp(torch.randn(4096, generator=g) * 6)       # wide dynamic range
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-30, 1.17549435e-38, 65504.0])
    p = kk.planes_split(x.cuda(), 3).float().cpu().double()
    rec = p[0] + p[1] + p[2]
    assert torch.equal(rec.float(), x), "hi + mid + lo must reproduce every fp32 value bit for bit"
    p1 = kk.planes_split(x.cuda(), 1).float().cpu()
    assert torch.equal(p1[0], x.to(torch.bfloat16).float())


CASES = [  # B, H, W, Cin, Cout, k, pad, groups
    (2, 14, 14, 64, 128, 3, 1, 1), (3, 9, 7, 128, 256, 3, 1, 2), (2, 12, 12, 32, 64, 3, 1, 1), (1, 8, 8, 256, 256, 1, 0, 1),
    (5, 7, 7, 64, 128, 1, 0, 1), (2, 20, 20, 128, 128, 3, 1, 2), (1, 30, 30, 64, 64, 3, 1, 1),
    (2, 56, 56, 64, 128, 3, 1, 2), (11, 7, 7, 128, 256, 3, 1, 2), (1, 5, 63, 32, 128, 3, 1, 1), (3, 28, 28, 64, 64, 3, 1, 1),
    # more tiles than CUs: the persistent kernel (tile 6) walks 2 tiles per workgroup, last round partial, ragged last tile
    (17, 55, 56, 128, 256, 3, 1, 2), (33, 28, 27, 256, 512, 1, 0, 1)]


@pytest.mark.parametrize("cfg", CASES)
def test_plane_convolution_matches_fp64(cfg):
    B, H, W, Cin, Cout, k, pad, groups = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg))
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)) + 0.2)
    w = torch.from_numpy(rng.standard_normal((Cout, Cin // groups, k, k)) / np.sqrt(Cin // groups * k * k))
    y_ref = F.conv2d(x, w, None, 1, pad, 1, groups)
    xd = nhwc(x)
    wd = w.float().permute(2, 3, 1, 0).contiguous().cuda()
    # yardsticks: the exact-fp32 MFMA kernel on the same fp32 operands
    y32 = kk.conv2d_fwd(xd, wd, None, None, 1, pad, groups)
    e32 = float((y32.permute(0, 3, 1, 2).cpu().double() - F.conv2d(xd.permute(0, 3, 1, 2).cpu().double(), wd.permute(3, 2, 0, 1).cpu().double(), None, 1, pad, 1, groups)).abs().max())
    xp = kk.planes_split(xd, 3)
    wf, wdg = kk.planes_split_weight(wd, groups, 3)
    y = None
    for t in [t for t in kk._plane_tiles(Cout // groups) if t != 6 or k * k * (Cin // groups // 32) >= 2]:   # every block tile: bit-identical outputs and statistics
        yt, (part, rows) = kk.conv2d_fwd_planes(xp, wf, k, k, 1, pad, groups, bn_stats=True, tile=t)
        st_t = part.sum(0)
        if y is not None:
            assert torch.equal(yt, y), t
        y = yt
    y_true = F.conv2d(xd.permute(0, 3, 1, 2).cpu().double(), wd.permute(3, 2, 0, 1).cpu().double(), None, 1, pad, 1, groups)
    e3 = float((y.permute(0, 3, 1, 2).cpu().double() - y_true).abs().max())
    sc = float(y_true.abs().max())
    if kk._halo_ok(k, k, 1, pad, H, W):      # resident-rows kernel: other summation order over K, same accuracy class
        yh, (ph, rh) = kk.conv2d_fwd_planes(xp, wf, k, k, 1, pad, groups, bn_stats=True, tile=5)
        eh = float((yh.permute(0, 3, 1, 2).cpu().double() - y_true).abs().max())
        print("fwd halo %s: |bf16x3 - fp64| %.3g (other tiles %.3g, fp32 MFMA %.3g)" % (cfg, eh, e3, e32))
        assert eh <= max(2.0 * e32, 4e-7 * sc), (eh, e32, sc)
        yhf = yh.double().reshape(-1, Cout).cpu().numpy()
        sth = ph.sum(0).cpu().numpy()
        assert rh == -(-yhf.shape[0] // 256)
        np.testing.assert_allclose(sth[:, 0], yhf.sum(0), rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(sth[:, 1], (yhf * yhf).sum(0), rtol=1e-10, atol=1e-8)
        assert torch.equal(kk.conv2d_fwd_planes(xp, wf, k, k, 1, pad, groups, tile=5), yh)      # run to run
    print("fwd %s: |bf16x3 - fp64| %.3g  |fp32 MFMA - fp64| %.3g  (scale %.3g)" % (cfg, e3, e32, sc))
    assert e3 <= max(2.0 * e32, 4e-7 * sc), (e3, e32, sc)
    yf = y.double().reshape(-1, Cout).cpu().numpy()
    st = part.sum(0).cpu().numpy()
    np.testing.assert_allclose(st[:, 0], yf.sum(0), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(st[:, 1], (yf * yf).sum(0), rtol=1e-10, atol=1e-8)
    # one plane: exact convolution of the bf16-rounded operands, fp32 accumulation
    y1 = kk.conv2d_fwd_planes(kk.planes_split(xd, 1), kk.planes_split_weight(wd, groups, 1)[0], k, k, 1, pad, groups, tile=1)
    rb = lambda t: t.float().to(torch.bfloat16).double()
    y1_ref = F.conv2d(rb(xd.permute(0, 3, 1, 2).cpu()), rb(wd.permute(3, 2, 0, 1).cpu()), None, 1, pad, 1, groups)
    np.testing.assert_allclose(y1.permute(0, 3, 1, 2).cpu().numpy(), y1_ref.numpy(), atol=2e-5 * sc, rtol=1e-5)
    if kk._halo_ok(k, k, 1, pad, H, W):
        y1h = kk.conv2d_fwd_planes(kk.planes_split(xd, 1), kk.planes_split_weight(wd, groups, 1)[0], k, k, 1, pad, groups, tile=5)
        np.testing.assert_allclose(y1h.permute(0, 3, 1, 2).cpu().numpy(), y1_ref.numpy(), atol=2e-5 * sc, rtol=1e-5)
    # input gradient
    if (Cin // groups) % 64 == 0:
        dy = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
        dyd = nhwc(dy)
        xr = x.clone().requires_grad_(True)
        dx_true = torch.autograd.grad(F.conv2d(xr, wd.permute(3, 2, 0, 1).cpu().double(), None, 1, pad, 1, groups), xr,
                                      dyd.permute(0, 3, 1, 2).cpu().double())[0]
        dx32 = kk.conv2d_dgrad(dyd, wd, tuple(xd.shape), None, 1, pad, groups)
        dx3 = kk.conv2d_dgrad_planes(kk.planes_split(dyd, 3), wdg, tuple(xd.shape), k, k, 1, pad, groups,
                                     tile=kk._plane_tiles(Cin // groups)[0])     # (tiles 0-4 are bit-identical; 5 below)
        for t in [t for t in kk._plane_tiles(Cin // groups) if t != 6 or k * k * (Cout // groups // 32) >= 2]:
            assert torch.equal(kk.conv2d_dgrad_planes(kk.planes_split(dyd, 3), wdg, tuple(xd.shape), k, k, 1, pad, groups,
                                                      tile=t), dx3), t
        ed32 = float((dx32.permute(0, 3, 1, 2).cpu().double() - dx_true).abs().max())
        if kk._halo_ok(k, k, 1, pad, H, W):
            dxh = kk.conv2d_dgrad_planes(kk.planes_split(dyd, 3), wdg, tuple(xd.shape), k, k, 1, pad, groups, tile=5)
            edh = float((dxh.permute(0, 3, 1, 2).cpu().double() - dx_true).abs().max())
            print("dgrad halo %s: |bf16x3 - fp64| %.3g (fp32 MFMA %.3g)" % (cfg, edh, ed32))
            assert edh <= max(2.0 * ed32, 4e-7 * float(dx_true.abs().max())), (edh, ed32)
        ed3 = float((dx3.permute(0, 3, 1, 2).cpu().double() - dx_true).abs().max())
        print("dgrad %s: |bf16x3 - fp64| %.3g  |fp32 MFMA - fp64| %.3g" % (cfg, ed3, ed32))
        assert ed3 <= max(2.0 * ed32, 4e-7 * float(dx_true.abs().max())), (ed3, ed32)


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 128, 3, 1, 1), (3, 9, 7, 128, 256, 3, 1, 2), (1, 8, 8, 256, 256, 1, 0, 1),
                                 (2, 20, 20, 128, 128, 3, 1, 2), (1, 30, 30, 64, 64, 3, 1, 1), (5, 7, 7, 128, 128, 3, 1, 1),
                                 (3, 28, 28, 128, 256, 3, 1, 2), (2, 56, 56, 128, 256, 3, 1, 2), (7, 14, 14, 256, 512, 3, 1, 2),
                                 (3, 33, 31, 64, 64, 3, 1, 1)])
def test_plane_weight_gradient_matches_fp64(cfg):
    """dW on planes: transposing LDS reads (ds_read_b64_tr_b16), split-K slabs, padded taps, ragged last chunk."""
    B, H, W, Cin, Cout, k, pad, groups = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg) + 1)
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)) + 0.2)
    dy = torch.from_numpy(rng.standard_normal((B, Cout, H, W)))
    xd, dyd = nhwc(x), nhwc(dy)
    w = torch.zeros(Cout, Cin // groups, k, k, dtype=torch.float64, requires_grad=True)
    dw_true = torch.autograd.grad(F.conv2d(xd.permute(0, 3, 1, 2).cpu().double(), w, None, 1, pad, 1, groups), w,
                                  dyd.permute(0, 3, 1, 2).cpu().double())[0]
    dw32 = torch.zeros(k, k, Cin // groups, Cout, device="cuda")
    kk.conv2d_wgrad(xd, dyd, dw32, 1, pad, groups)
    dw3 = torch.full((k, k, Cin // groups, Cout), float("nan"), device="cuda")
    kk.conv2d_wgrad_planes(kk.planes_split(xd, 3), kk.planes_split(dyd, 3), dw3, pad, groups)
    e32 = float((dw32.permute(3, 2, 0, 1).cpu().double() - dw_true).abs().max())
    e3 = float((dw3.permute(3, 2, 0, 1).cpu().double() - dw_true).abs().max())
    sc = float(dw_true.abs().max())
    print("wgrad %s: |bf16x3 - fp64| %.3g  |fp32 MFMA - fp64| %.3g  (scale %.3g)" % (cfg, e3, e32, sc))
    assert e3 <= max(2.0 * e32, 4e-7 * sc), (e3, e32, sc)
    dw3b = torch.zeros_like(dw3)
    kk.conv2d_wgrad_planes(kk.planes_split(xd, 3), kk.planes_split(dyd, 3), dw3b, pad, groups)
    assert torch.equal(dw3, dw3b)                                   # deterministic
    # every (tile, split-K) plan the autotuner may choose: same accuracy, each deterministic
    key = ("pwgrad", 3, B, H, W, Cin, Cout, k, k, pad, groups)
    try:
        for plan in kk._PWGRAD_PLANS:
            if plan >= 64 and not (k == 3 and pad == 1 and W <= 63):
                continue                                         # (tap-fused plans: 3x3 / pad 1 layers)
            kk._tile_cache[key] = plan
            dwp = torch.full_like(dw3, float("nan"))
            kk.conv2d_wgrad_planes(kk.planes_split(xd, 3), kk.planes_split(dyd, 3), dwp, pad, groups)
            ep = float((dwp.permute(3, 2, 0, 1).cpu().double() - dw_true).abs().max())
            assert ep <= max(2.0 * e32, 4e-7 * sc), (plan, ep, e32, sc)
    finally:
        kk._tile_cache.pop(key, None)
