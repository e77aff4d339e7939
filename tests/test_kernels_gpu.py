"""Kernel-level parity (-m gpu): every HIP op called through the C ABI vs torch-CPU maths (fp64 where cheap).
fp32 tolerances are stated per test; they follow fp32 accumulation over the contraction length."""
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


def from_nhwc(t):
    return t.cpu().permute(0, 3, 1, 2).double()


def to_hwio(w):
    return w.permute(2, 3, 1, 0).contiguous().float().cuda()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, p, g
    (2, 14, 14, 64, 128, 3, 1, 1, 1),
    (2, 14, 14, 64, 128, 3, 1, 1, 2),      # ResNeSt radix conv (groups = 2)
    (3, 7, 7, 256, 64, 1, 1, 0, 1),        # ragged M (147 rows)
    (2, 16, 16, 64, 64, 3, 2, 1, 1),       # resnet18 strided 3x3
    (2, 16, 16, 64, 128, 1, 2, 0, 1),      # resnet18 downsample_conv
    (2, 20, 20, 32, 32, 3, 1, 1, 1),       # stem 32->32 (N = 32 tile)
    (4, 28, 28, 128, 512, 1, 1, 0, 1),     # wide N, several M tiles
    (1, 9, 9, 2048, 64, 1, 1, 0, 1),       # head conv1x1 (K = 2048)
    (2, 12, 12, 512, 1024, 3, 1, 1, 2),    # layer4 radix conv
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    B, H, W, Cin, Cout, k, s, p, g = case
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)))
    w = torch.from_numpy(rng.standard_normal((Cout, Cin // g, k, k)) / np.sqrt(Cin // g * k * k))
    bias = torch.from_numpy(rng.standard_normal(Cout))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, s, p, 1, g)
    add = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
    dy = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
    y_ref.backward(dy)
    kk = K()
    y = kk.conv2d_fwd(nhwc(x), to_hwio(w), bias.float().cuda(), None, s, p, g, False)
    tol = 2e-5 * np.sqrt(Cin // g * k * k)
    np.testing.assert_allclose(from_nhwc(y).numpy(), y_ref.detach().numpy(), atol=tol, rtol=1e-5)
    y2 = kk.conv2d_fwd(nhwc(x), to_hwio(w), None, nhwc(add), s, p, g, True)
    np.testing.assert_allclose(from_nhwc(y2).numpy(), torch.relu(F.conv2d(x, w, None, s, p, 1, g) + add).numpy(),
                               atol=tol, rtol=1e-5)
    dx = kk.conv2d_dgrad(nhwc(dy), to_hwio(w), (B, H, W, Cin), None, s, p, g)
    np.testing.assert_allclose(from_nhwc(dx).numpy(), xr.grad.numpy(), atol=2e-5 * np.sqrt(Cout // g * k * k), rtol=1e-5)
    dw = torch.empty((k, k, Cin // g, Cout), dtype=torch.float32, device="cuda")
    kk.conv2d_wgrad(nhwc(x), nhwc(dy), dw, s, p, g)
    dw_ref = wr.grad.permute(2, 3, 1, 0).numpy()
    np.testing.assert_allclose(dw.cpu().double().numpy(), dw_ref, atol=3e-5 * np.sqrt(B * y_ref.shape[2] * y_ref.shape[3]),
                               rtol=1e-5)


def test_wgrad_split_reduction_large_m():
    """Many pixels -> several split-K slabs + the deterministic slab reduction."""
    B, H, W, Cin, Cout = 8, 56, 56, 32, 64
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((B, Cout, H, W)).astype(np.float32))
    kk = K()
    dw = torch.empty((3, 3, Cin, Cout), dtype=torch.float32, device="cuda")
    kk.conv2d_wgrad(nhwc(x), nhwc(dy), dw, 1, 1, 1)
    dw2 = torch.empty_like(dw)
    kk.conv2d_wgrad(nhwc(x), nhwc(dy), dw2, 1, 1, 1)
    assert torch.equal(dw, dw2), "wgrad must be deterministic"
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), 1, 1).permute(2, 3, 1, 0)
    np.testing.assert_allclose(dw.cpu().double().numpy(), ref.numpy(), atol=5e-3, rtol=1e-4)


def test_stem_im2col_path():
    """3-channel 3x3/2 stem as im2col (K 27 -> 32) + 1x1 conv; also its weight gradient."""
    rng = np.random.default_rng(3)
    B, Cin, H, Cout, k, s, p = 3, 3, 34, 32, 3, 2, 1
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, H)))
    w = torch.from_numpy(rng.standard_normal((Cout, Cin, k, k)) * 0.2)
    kk = K()
    col = kk.im2col_nchw(x.float().cuda(), k, s, p, 32)
    wflat = to_hwio(w).reshape(-1)
    wpad = kk.pad_rows(wflat, k * k * Cin * Cout, 32 * Cout).view(1, 1, 32, Cout)
    y = kk.conv2d_fwd(col, wpad)
    ref = F.conv2d(x, w, None, s, p)
    np.testing.assert_allclose(from_nhwc(y).numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
    dy = torch.from_numpy(rng.standard_normal(tuple(ref.shape)))
    dw = torch.empty((1, 1, 32, Cout), dtype=torch.float32, device="cuda")
    kk.conv2d_wgrad(col, nhwc(dy), dw)
    dref = torch.nn.grad.conv2d_weight(x, tuple(w.shape), dy, s, p).permute(2, 3, 1, 0).reshape(27, Cout)
    np.testing.assert_allclose(dw.view(32, Cout)[:27].cpu().double().numpy(), dref.numpy(), atol=5e-4, rtol=1e-5)
    assert float(dw.view(32, Cout)[27:].abs().max()) == 0.0


@pytest.mark.parametrize("shape,relu,res", [((4, 14, 14, 64), True, False), ((3, 7, 7, 2048), False, True),
                                            ((6, 1, 1, 32), True, False), ((2, 28, 28, 32), True, True),
                                            ((70, 1, 1, 256), True, False), ((9, 1, 1, 64), False, True),   # one-launch backward
                                            # more small shapes: whole rows per pass / 1024-channel slabs / ragged
                                            ((70, 1, 1, 128), True, False), ((70, 1, 1, 2048), True, False),
                                            ((5, 2, 2, 128), True, False), ((3, 1, 1, 96), True, False)])
def test_bn_fwd_bwd(shape, relu, res):
    rng = np.random.default_rng(4)
    B, H, W, C = shape
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)) * 2 + 0.5)
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, C))
    beta = torch.from_numpy(rng.standard_normal(C) * 0.1)
    rm = torch.from_numpy(rng.standard_normal(C) * 0.1)
    rv = torch.from_numpy(rng.uniform(0.5, 1.5, C))
    r = torch.from_numpy(rng.standard_normal((B, C, H, W))) if res else None
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if res:
        y_ref = y_ref + r
    if relu:
        y_ref = torch.relu(y_ref)
    dy = torch.from_numpy(rng.standard_normal((B, C, H, W)))
    y_ref.backward(dy)
    kk = K()
    rm_d, rv_d = rm.float().cuda(), rv.float().cuda()
    xd = nhwc(x)
    y, saved = kk.bn_fwd(xd, gamma.float().cuda(), beta.float().cuda(), rm_d, rv_d, True, relu, nhwc(r) if res else None)
    np.testing.assert_allclose(from_nhwc(y).numpy(), y_ref.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm_ref.numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv_ref.numpy(), atol=1e-6, rtol=1e-5)
    dg = torch.empty(C, dtype=torch.float32, device="cuda")
    db = torch.empty(C, dtype=torch.float32, device="cuda")
    dx, gout = kk.bn_bwd(nhwc(dy), y if relu else None, xd, saved, True, dg, db, want_gout=True)
    if relu:      # the 1-bit sign mask written by the forward gives the same backward as the activation itself
        y2, saved2, bits = kk.bn_fwd(xd, gamma.float().cuda(), beta.float().cuda(), rm.float().cuda(), rv.float().cuda(),
                                     True, True, nhwc(r) if res else None, want_mask=True)
        assert torch.equal(y2, y)
        dg2, db2 = torch.empty_like(dg), torch.empty_like(db)
        dx2, gout2 = kk.bn_bwd(nhwc(dy), None, xd, saved2, True, dg2, db2, want_gout=True, mask=bits)
        assert torch.equal(dx2, dx) and torch.equal(gout2, gout) and torch.equal(dg2, dg) and torch.equal(db2, db)
    sc = float(xr.grad.abs().max())
    np.testing.assert_allclose(from_nhwc(dx).numpy(), xr.grad.numpy(), atol=3e-5 * max(sc, 1), rtol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), atol=1e-4 * max(float(gr.grad.abs().max()), 1), rtol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), atol=1e-4 * max(float(br.grad.abs().max()), 1), rtol=1e-4)
    mask = (y_ref.detach() > 0).double() if relu else 1.0
    np.testing.assert_allclose(from_nhwc(gout).numpy(), (dy * mask).numpy(), atol=1e-6)
    # eval mode uses the running statistics
    y_e, _ = kk.bn_fwd(xd, gamma.float().cuda(), beta.float().cuda(), rm_d, rv_d, False, False)
    ref_e = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5)
    np.testing.assert_allclose(from_nhwc(y_e).numpy(), ref_e.numpy(), atol=2e-5, rtol=1e-5)


def test_pools():
    rng = np.random.default_rng(5)
    kk = K()
    x = torch.from_numpy(rng.standard_normal((2, 32, 17, 17)))
    dy_of = lambda y: torch.from_numpy(rng.standard_normal(tuple(y.shape)))
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    dy = dy_of(y_ref)
    y_ref.backward(dy)
    y, arg = kk.maxpool_fwd(nhwc(x))
    np.testing.assert_array_equal(from_nhwc(y).numpy(), y_ref.detach().float().double().numpy())
    dx = kk.maxpool_bwd(nhwc(dy), arg, (2, 17, 17, 32))
    np.testing.assert_allclose(from_nhwc(dx).numpy(), xr.grad.numpy(), atol=1e-6)
    for (k, s, p, ceil, cip) in [(3, 2, 1, False, True), (2, 2, 0, True, False), (3, 1, 1, False, True)]:
        xr = x.clone().requires_grad_(True)
        y_ref = F.avg_pool2d(xr, k, s, p, ceil_mode=ceil, count_include_pad=cip)
        dy = dy_of(y_ref)
        y_ref.backward(dy)
        y = kk.avgpool_fwd(nhwc(x), k, s, p, ceil, cip)
        assert tuple(y.shape[1:3]) == tuple(y_ref.shape[2:])
        np.testing.assert_allclose(from_nhwc(y).numpy(), y_ref.detach().numpy(), atol=1e-6)
        dx = kk.avgpool_bwd(nhwc(dy), (2, 17, 17, 32), k, s, p, ceil, cip)
        np.testing.assert_allclose(from_nhwc(dx).numpy(), xr.grad.numpy(), atol=1e-6)
    t = kk.nhwc_to_nchw(nhwc(x))
    np.testing.assert_array_equal(t.cpu().numpy(), x.float().numpy())
    np.testing.assert_array_equal(kk.nchw_to_nhwc(x.float().cuda()).cpu().numpy(), nhwc(x).cpu().numpy())


def test_split_attention_glue():
    rng = np.random.default_rng(6)
    kk = K()
    B, H, W, Cp = 3, 10, 10, 64
    x = torch.from_numpy(rng.standard_normal((B, H, W, 2 * Cp)))
    z = torch.from_numpy(rng.standard_normal((B, 2 * Cp)))
    dout = torch.from_numpy(rng.standard_normal((B, H, W, Cp)))
    xd = x.float().cuda()
    gap = kk.sa_gap(xd)
    np.testing.assert_allclose(gap.cpu().numpy(), (x[..., :Cp] + x[..., Cp:]).mean((1, 2)).numpy(), atol=1e-6)
    a = kk.radix_softmax_fwd(z.float().cuda())
    a_ref = torch.softmax(z.view(B, 2, Cp), 1).view(B, 2 * Cp)
    np.testing.assert_allclose(a.cpu().numpy(), a_ref.numpy(), atol=1e-6)
    out = kk.sa_apply_fwd(xd, a)
    out_ref = x[..., :Cp] * a_ref[:, None, None, :Cp] + x[..., Cp:] * a_ref[:, None, None, Cp:]
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.numpy(), atol=1e-5)
    da = kk.sa_dattn(xd, dout.float().cuda())
    da_ref = torch.cat([(dout * x[..., :Cp]).sum((1, 2)), (dout * x[..., Cp:]).sum((1, 2))], 1)
    np.testing.assert_allclose(da.cpu().numpy(), da_ref.numpy(), atol=1e-4, rtol=1e-5)
    dz = kk.radix_softmax_bwd(a, da)
    zr = z.clone().requires_grad_(True)
    torch.softmax(zr.view(B, 2, Cp), 1).view(B, 2 * Cp).backward(da_ref)
    np.testing.assert_allclose(dz.cpu().numpy(), zr.grad.numpy(), atol=1e-4, rtol=1e-4)
    dgap = torch.from_numpy(rng.standard_normal((B, Cp)))
    dx = kk.sa_apply_bwd(dout.float().cuda(), a, dgap.float().cuda())
    dx_ref = torch.cat([dout * a_ref[:, None, None, :Cp], dout * a_ref[:, None, None, Cp:]], -1) + \
        torch.cat([dgap, dgap], 1)[:, None, None, :] / (H * W)
    np.testing.assert_allclose(dx.cpu().numpy(), dx_ref.numpy(), atol=1e-5)


@pytest.mark.parametrize("shape,training", [((3, 10, 10, 64), True), ((2, 7, 9, 128), True), ((5, 4, 4, 32), False)])
def test_split_attention_fused_with_its_batchnorm(shape, training):
    """bn0 + ReLU folded into the split-attention passes (the activation relu(bn0(x0)) is never stored): GAP, weighted
    radix sum, d(attention), and the fused backward [d(h0) -> ReLU mask -> BatchNorm backward] against fp64 autograd of
    the unfused chain (timm/models/layers/split_attn.py:62-80 + bn0 / act0)."""
    rng = np.random.default_rng(sum(shape))
    kk = K()
    B, H, W, Cp = shape
    C2 = 2 * Cp
    x0 = torch.from_numpy(rng.standard_normal((B, H, W, C2)) * 1.5 + 0.4)
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, C2)); beta = torch.from_numpy(rng.standard_normal(C2) * 0.3)
    rm0 = torch.from_numpy(rng.standard_normal(C2) * 0.2); rv0 = torch.from_numpy(rng.uniform(0.5, 2.0, C2))
    a = torch.softmax(torch.from_numpy(rng.standard_normal((B, 2, Cp))), 1).reshape(B, C2)
    dout = torch.from_numpy(rng.standard_normal((B, H, W, Cp)))
    dgap = torch.from_numpy(rng.standard_normal((B, Cp)))
    # fp64 reference of the unfused chain
    xr = x0.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    h = torch.relu(F.batch_norm(xr.permute(0, 3, 1, 2), rm, rv, gr, br, training, 0.1, 1e-5)).permute(0, 2, 3, 1)
    gap_ref = (h[..., :Cp] + h[..., Cp:]).mean((1, 2))
    out_ref = h[..., :Cp] * a[:, None, None, :Cp] + h[..., Cp:] * a[:, None, None, Cp:]
    da_ref = torch.cat([(dout * h[..., :Cp]).sum((1, 2)), (dout * h[..., Cp:]).sum((1, 2))], 1).detach()
    ((out_ref * dout).sum() + (gap_ref * dgap).sum()).backward()
    # HIP: statistics only, then the fused passes
    f = lambda t: t.float().cuda()
    rmd, rvd = f(rm0), f(rv0)
    xd = f(x0)
    saved = kk.bn_stats(xd, f(gamma), f(beta), rmd, rvd, training)
    np.testing.assert_allclose(kk.sa_gap(xd, saved).cpu().numpy(), gap_ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(kk.sa_apply_fwd(xd, f(a), saved).cpu().numpy(), out_ref.detach().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(kk.sa_dattn(xd, f(dout), saved).cpu().numpy(), da_ref.numpy(), atol=2e-4, rtol=1e-5)
    if training:
        np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), atol=1e-6, rtol=1e-5)
    # the apply pass alone reproduces the activation (and thereby the sign pattern) the fused kernels see
    np.testing.assert_allclose(kk.bn_apply(xd, saved, True).cpu().numpy(), h.detach().numpy(), atol=1e-5, rtol=1e-5)
    dg, db = torch.zeros(C2, device="cuda"), torch.zeros(C2, device="cuda")
    dx = kk.sa_bn_bwd(f(dout), f(a), f(dgap), xd, saved, training, dg, db)
    sc = float(xr.grad.abs().max())
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=2e-6 * max(sc, 1.0), rtol=2e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), atol=2e-5 * float(gr.grad.abs().max()), rtol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), atol=2e-5 * float(br.grad.abs().max()), rtol=1e-5)
    # the same backward from the per-image statistics of the d(attention) pass (no reduction pass over dout / x0):
    # g is affine in (a, dgap), so the BatchNorm sums follow from S1..S4 -- equal to the reduced ones to fp32 rounding
    da2, sums = kk.sa_dattn(xd, f(dout), saved, want_stats=True)
    assert torch.equal(da2, kk.sa_dattn(xd, f(dout), saved))
    dg2, db2 = torch.zeros(C2, device="cuda"), torch.zeros(C2, device="cuda")
    dx2 = kk.sa_bn_bwd(f(dout), f(a), f(dgap), xd, saved, training, dg2, db2, sums=sums)
    for got, ref in ((dx2, dx), (dg2, dg), (db2, db)):
        scl = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 3e-6 * scl + 1e-7, (float((got - ref).abs().max()), scl)


def test_colsum_relu_axpby_matmul_tn():
    rng = np.random.default_rng(7)
    kk = K()
    a = torch.from_numpy(rng.standard_normal((37, 640)).astype(np.float32))
    out = torch.empty(640, dtype=torch.float32, device="cuda")
    kk.colsum(a.cuda(), out)
    np.testing.assert_allclose(out.cpu().numpy(), a.double().sum(0).numpy(), atol=1e-5)
    y = torch.from_numpy(rng.standard_normal((37, 640)).astype(np.float32))
    np.testing.assert_array_equal(kk.relu_bwd(a.cuda(), y.cuda()).cpu().numpy(), (a * (y > 0)).numpy())
    np.testing.assert_allclose(kk.axpby(a.cuda(), y.cuda(), 2.0, -0.5).cpu().numpy(), (2 * a - 0.5 * y).numpy(), atol=1e-6)
    p = torch.from_numpy(rng.standard_normal((700, 192)).astype(np.float32))
    q = torch.from_numpy(rng.standard_normal((700, 64)).astype(np.float32))
    o = torch.empty((192, 64), dtype=torch.float32, device="cuda")
    kk.matmul_tn(p.cuda(), q.cuda(), o)
    np.testing.assert_allclose(o.cpu().numpy(), (p.double().t() @ q.double()).numpy(), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("h,w", [(7, 7), (9, 9), (1, 1), (3, 5)])
def test_posenc(h, w):
    from oracle import torch_oracle as O
    pe = K().posenc_sine(h, w, 64, torch.device("cuda"))
    ref = O.posenc_sine(h, w, 64).reshape(64, h * w).t()
    np.testing.assert_allclose(pe.cpu().numpy(), ref.numpy(), atol=2e-6)


def test_loss_fwd_bwd():
    rng = np.random.default_rng(8)
    kk = K()
    B, C, S, N, lam, power = 9, 10, 10, 49, 0.7, 2
    logits = torch.from_numpy(rng.standard_normal((B, C)) * 3)
    y = torch.from_numpy(rng.integers(0, C, B))
    area_part = torch.from_numpy(rng.uniform(100, 200, B))
    lr = logits.clone().requires_grad_(True)
    ar = area_part.clone().requires_grad_(True)
    logp = F.log_softmax(lr, 1)
    nll = F.nll_loss(logp, y)
    term = (ar.sum() / (B * S * N)) ** power
    loss = nll + lam * term
    loss.backward()
    lp, stats = kk.slot_loss_fwd(logits.float().cuda(), y.cuda(), area_part.float().cuda(), B * S * N, lam, power)
    np.testing.assert_allclose(lp.cpu().numpy(), logp.detach().numpy(), atol=2e-6)
    acc = float((logits.argmax(1) == y).double().mean())
    np.testing.assert_allclose(stats[:4].cpu().numpy(), [float(loss), float(nll), float(term), acc], rtol=2e-6, atol=1e-6)
    one = torch.ones(1, dtype=torch.float32, device="cuda")
    dl, ga = kk.slot_loss_bwd(lp, y.cuda(), stats, one, None, None, None, B * S * N, lam, power)
    np.testing.assert_allclose(dl.cpu().numpy(), lr.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(float(ga), float(ar.grad[0]), rtol=1e-5)
    assert float(stats[5]) == 0.0


def test_loss_label_out_of_range_is_flagged_not_dereferenced():
    """F.nll_loss raises on a target outside [0, C) (reference slot_model.py:121); here the kernels never index with
    it: loss / nll become NaN, the row's dlogits NaN, stats[5] counts the offenders and engine.calculation raises."""
    kk = K()
    B, C = 6, 5
    logits = torch.randn(B, C, device="cuda")
    y = torch.tensor([0, 4, 5, 2, -1, 1], device="cuda")
    lp, stats = kk.slot_loss_fwd(logits, y, None, 1.0, 0.0, 1.0)
    assert float(stats[5]) == 2.0 and bool(torch.isnan(stats[0])) and bool(torch.isnan(stats[1]))
    np.testing.assert_allclose(lp.cpu().numpy(), F.log_softmax(logits.cpu(), 1).numpy(), atol=2e-6)
    one = torch.ones(1, dtype=torch.float32, device="cuda")
    dl, _ = kk.slot_loss_bwd(lp, y, stats, one, None, None, None, 1.0, 0.0, 1.0)
    bad = torch.isnan(dl).all(dim=1).cpu().tolist()
    assert bad == [False, False, True, False, True, False]
    from scouter_amd.engine import _DeviceMeter
    meter = _DeviceMeter()
    meter.add_device(stats)
    with pytest.raises(IndexError, match="out of bounds"):
        meter.read()


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("case", [(3, 14, 14, 64, 128, 3, 1, 1, 2), (2, 9, 9, 256, 64, 1, 1, 0, 1), (5, 1, 1, 64, 32, 1, 1, 0, 1),
                                  (2, 30, 30, 32, 32, 3, 1, 1, 1), (3, 17, 13, 128, 256, 1, 1, 0, 1)])
def test_conv_fused_bn_statistics(case, with_bias):
    """conv epilogue -> per-tile fp64 channel sums -> bn_fwd(stats=...) == bn_fwd computing its own statistics.
    with_bias: the row-major path; without (every conv in front of a BatchNorm in the backbones): lane-local fp64
    column partials taken straight from the MFMA accumulators -- incl. partial last M tiles and a mean of ~6 sigma."""
    B, H, W, Cin, Cout, k, s, p, g = case
    rng = np.random.default_rng(11)
    x = nhwc(torch.from_numpy(rng.standard_normal((B, Cin, H, W)) + 0.3))
    w = to_hwio(torch.from_numpy(rng.standard_normal((Cout, Cin // g, k, k)) / np.sqrt(Cin // g * k * k) + 0.02))
    bias = torch.from_numpy(rng.standard_normal(Cout)).float().cuda() if with_bias else None
    kk = K()
    y_plain = kk.conv2d_fwd(x, w, bias, None, s, p, g)
    y, (part, rows) = kk.conv2d_fwd(x, w, bias, None, s, p, g, False, bn_stats=True)
    assert torch.equal(y, y_plain)
    M = y.numel() // Cout
    sums = part.sum(0).cpu().numpy()
    yd = y.double().view(M, Cout).cpu().numpy()
    np.testing.assert_allclose(sums[:, 0], yd.sum(0), rtol=1e-10, atol=1e-8)         # fp64 from the first addition on
    np.testing.assert_allclose(sums[:, 1], (yd * yd).sum(0), rtol=1e-10, atol=1e-8)
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, Cout)).float().cuda()
    beta = torch.from_numpy(rng.standard_normal(Cout)).float().cuda()
    outs = []
    for st in (None, (part, rows)):
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        o, saved = kk.bn_fwd(y, gamma, beta, rm, rv, True, True, stats=st)
        outs.append((o, saved, rm, rv))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-6)


def _bf16_round(t):
    return t.float().to(torch.bfloat16).double()


@pytest.mark.parametrize("cfg", [
    # B, H, W, Cin, Cout, k, stride, pad, groups
    (2, 14, 14, 64, 128, 3, 1, 1, 2), (3, 9, 7, 128, 64, 1, 1, 0, 1), (2, 12, 12, 32, 32, 3, 1, 1, 1),
    (1, 8, 8, 256, 256, 3, 1, 1, 2), (2, 10, 10, 64, 64, 3, 2, 1, 1), (5, 7, 7, 192, 96, 1, 1, 0, 1),
    (3, 14, 14, 128, 128, 3, 1, 1, 1), (2, 28, 28, 64, 128, 3, 1, 1, 1), (7, 7, 7, 128, 256, 1, 1, 0, 1),
    (2, 70, 70, 64, 64, 3, 1, 1, 1),
    # 32-channel groups: the weight gradient's ragged 64-wide tiles (stem / first grouped layer of ResNeSt; 96 = 64 + 32)
    (3, 17, 19, 64, 128, 3, 1, 1, 2), (2, 23, 21, 32, 64, 3, 1, 1, 1), (5, 9, 11, 96, 32, 1, 1, 0, 1),
    (4, 30, 30, 64, 192, 3, 1, 1, 2)])
def test_conv_bf16_inputs_fp32_accumulate(cfg):
    """bf16 mode: the kernels must equal an exact convolution of the bf16-ROUNDED operands (fp32 accumulation error
    only), for the forward (+ fused BN statistics) and the stride-1 input gradient."""
    B, H, W, Cin, Cout, k, stride, pad, groups = cfg
    rng = np.random.default_rng(sum(cfg))
    x = torch.from_numpy(rng.standard_normal((B, Cin, H, W)))
    w = torch.from_numpy(rng.standard_normal((Cout, Cin // groups, k, k)) * 0.1)
    kk = K()
    old_min = kk.BF16_MIN_PIXELS
    kk.BF16_MIN_PIXELS = 1                                # (the model keeps layers under 1024 pixels in fp32)
    try:
        y_ref = F.conv2d(_bf16_round(x), _bf16_round(w), None, stride, pad, 1, groups)
        xd, wd = nhwc(x), w.float().permute(2, 3, 1, 0).contiguous().cuda()
        y, (part, rows) = kk.conv2d_fwd(xd, wd, None, None, stride, pad, groups, False, bn_stats=True, precision="bf16")
        sc = float(y_ref.abs().max())
        np.testing.assert_allclose(from_nhwc(y).numpy(), y_ref.numpy(), atol=2e-5 * sc, rtol=1e-5)
        st = part.sum(0).cpu().numpy()                      # fused statistics of what was written
        yf = y.double().reshape(-1, Cout).cpu().numpy()
        np.testing.assert_allclose(st[:, 0], yf.sum(0), rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(st[:, 1], (yf * yf).sum(0), rtol=1e-9, atol=1e-6)
        if stride == 1:
            dy = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
            xr = _bf16_round(x).requires_grad_(True)        # only its shape matters for the input gradient
            dx_ref = torch.autograd.grad(F.conv2d(xr, _bf16_round(w), None, stride, pad, 1, groups), xr, _bf16_round(dy))[0]
            dx = kk.conv2d_dgrad(nhwc(dy), wd, tuple(xd.shape), None, stride, pad, groups, precision="bf16")
            np.testing.assert_allclose(from_nhwc(dx).numpy(), dx_ref.numpy(), atol=2e-5 * float(dx_ref.abs().max()), rtol=1e-5)
        # weight gradient: bf16 kernel where it applies (same-size, 32-multiples per group), fp32 kernel elsewhere
        dy = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
        uses_bf16 = stride == 1 and (k == 1 or 64 // W + 1 < H)
        xs, dys = (_bf16_round(x), _bf16_round(dy)) if uses_bf16 else (x, dy)
        wr = w.clone().requires_grad_(True)
        dw_ref = torch.autograd.grad(F.conv2d(xs, wr, None, stride, pad, 1, groups), wr, dys)[0]
        dw = torch.zeros_like(wd)
        kk.conv2d_wgrad(xd, nhwc(dy), dw, stride, pad, groups, precision="bf16")
        np.testing.assert_allclose(dw.permute(3, 2, 0, 1).cpu().numpy(), dw_ref.numpy(),
                                   atol=3e-5 * float(dw_ref.abs().max()), rtol=1e-5)
    finally:
        kk.BF16_MIN_PIXELS = old_min


@pytest.mark.parametrize("precision", ["fp32", "bf16", "planes"])
@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, pad, groups, two BatchNorms?, shortcut addend?
    (3, 14, 14, 256, 64, 1, 0, 1, True, True),       # conv1 of a bottleneck: finishes bn3 + the downsample BatchNorm
    (2, 9, 9, 128, 128, 3, 1, 2, False, False),      # radix conv: finishes bn1 (ragged last M tile)
    (5, 7, 7, 64, 32, 1, 0, 1, False, True),
    (9, 12, 12, 64, 64, 3, 1, 1, True, False),
    (45, 28, 27, 128, 128, 3, 1, 2, True, True),      # > 256 plane tiles: the persistent kernel's second round
    (7, 13, 11, 256, 128, 1, 0, 1, True, True),       # persistent fused kernel: K = 128, two column groups, ragged M
    (3, 9, 9, 512, 256, 1, 0, 1, False, True),        # K = 256: 32 x 32 wave tiles, eight column groups
    (11, 28, 28, 64, 64, 1, 0, 1, False, False)])     # K = 64, N = 64: one column group, no shortcut gradient
def test_input_gradient_epilogue_finishes_batchnorm_backward(case, precision, monkeypatch):
    """conv dgrad with a BnBwdFuse == conv dgrad, then ReLU mask, then the BatchNorm backward's own reduction: the masked
    gradient bit for bit (every tile), the fp64 partial sums to 1e-10, dx / dgamma / dbeta of the BatchNorm(s) to fp32
    rounding -- for the fp32-MFMA, the bf16-input and the bf16x3 plane kernels."""
    B, H, W, Cin, Cout, k, pad, g, two, with_add = case
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)         # every producer class (also the model default)
    rng = np.random.default_rng(sum(case[:8]))
    if precision == "planes" and ((Cin // g) % 64 or (Cout // g) % 32):
        pytest.skip("plane input gradient needs 64-multiples of input channels per group")
    rnd = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()
    dy = rnd(B, H, W, Cout)
    w = (rnd(k, k, Cin // g, Cout) / np.sqrt(Cout * k * k)).contiguous()
    addend = rnd(B, H, W, Cin) if with_add else None
    # the BatchNorm(s) whose output gradient the convolution produces: forward first (mask, saved statistics)
    bns = []
    res = rnd(B, H, W, Cin)
    for i in range(2 if two else 1):
        x = rnd(B, H, W, Cin) * 1.7 + 0.4
        gamma, beta = torch.rand(Cin, device="cuda") + 0.5, rnd(Cin)
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        bns.append((x, gamma, beta, rm, rv))
    x1, g1, b1, rm1, rv1 = bns[0]
    if two:                                            # out = relu(bn1(x1) + bn2(x2))
        x2, g2, b2, rm2, rv2 = bns[1]
        y2, saved2 = kk.bn_fwd(x2, g2, b2, rm2, rv2, True, False)
        _, saved1, mask = kk.bn_fwd(x1, g1, b1, rm1, rv1, True, True, residual=y2, want_mask=True)
    else:                                              # out = relu(bn1(x1) + shortcut)
        _, saved1, mask = kk.bn_fwd(x1, g1, b1, rm1, rv1, True, True, residual=res, want_mask=True)

    def dgrad(post, tile=None):
        if precision == "planes":
            dyp = kk.planes_split(dy, 3)
            _, wdg = kk.planes_split_weight(w, g, 3)
            return kk.conv2d_dgrad_planes(dyp, wdg, (B, H, W, Cin), k, k, 1, pad, g, addend, tile=tile, post=post)
        if tile is not None:
            b16 = precision == "bf16" and B * H * W >= kk.BF16_MIN_PIXELS
            kk._tile_cache[("dgrad", b16, B, H, W, Cin, Cout, k, k, 1, pad, g)] = tile
            kk._tile_cache[("dgrad+bn", 2 if two else 1, addend is not None, b16, B, H, W, Cin, Cout, k, k, 1, pad, g)] = tile
        return kk.conv2d_dgrad(dy, w, (B, H, W, Cin), addend, 1, pad, g, precision=precision, post=post)

    tiles = (kk._plane_tiles(Cin // g, 3, kk._halo_ok(k, k, 1, pad, H, W, 2)) if precision == "planes"
             else [t for t in range(4) if kk._tile_legal(Cin // g, t)])
    if precision == "fp32" and kk._pw_persist_legal(B * H * W, Cout, Cin, k, k, 1, pad, g, True):
        tiles = list(tiles) + [4]              # the persistent pointwise kernel carries the fused epilogue too
    if precision == "planes" and k * k * (Cout // g // 32) < 2:
        tiles = [t for t in tiles if t != 6]              # (the persistent tile needs two K-tiles)
    for t in tiles:
        # unfused chain on the SAME tile (plane tile 5 sums K in another order than tiles 0-4)
        d_plain = dgrad(None, t)
        dg = [torch.zeros(Cin, device="cuda") for _ in range(4)]
        dx1_ref, gout = kk.bn_bwd(d_plain, None, x1, saved1, True, dg[0], dg[1], True, mask=mask)
        if two:
            dx2_ref, _ = kk.bn_bwd(gout, None, x2, saved2, True, dg[2], dg[3])
        post = kk.BnBwdFuse(mask, [(x1, saved1)] + ([(x2, saved2)] if two else []))
        gf = dgrad(post, t)
        assert post.applied
        assert torch.equal(gf, gout), (t, float((gf - gout).abs().max()))
        gd = gout.double().view(-1, Cin)
        for i, (x, saved) in enumerate(post.entries):
            part, rows = post.ext(i)
            sums = part.sum(0).cpu().numpy()
            xhat = ((x.view(-1, Cin) - saved[0]) * saved[1]).double()          # fp32 xhat, as every kernel forms it
            np.testing.assert_allclose(sums[:, 0], gd.sum(0).cpu().numpy(), rtol=1e-10, atol=1e-9)
            np.testing.assert_allclose(sums[:, 1], (gd * xhat).sum(0).cpu().numpy(), rtol=1e-10, atol=1e-9)
        fg = [torch.zeros(Cin, device="cuda") for _ in range(4)]
        dx1, gg = kk.bn_bwd(gf, None, x1, saved1, True, fg[0], fg[1], True, ext=post.ext(0))
        assert gg is gf
        outs, refs = [dx1, fg[0], fg[1]], [dx1_ref, dg[0], dg[1]]
        if two:
            dx2, _ = kk.bn_bwd(gf, None, x2, saved2, True, fg[2], fg[3], ext=post.ext(1))
            outs += [dx2, fg[2], fg[3]]
            refs += [dx2_ref, dg[2], dg[3]]
        for a, b in zip(outs, refs):
            sc = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-6 * sc + 1e-7, (t, float((a - b).abs().max()), sc)
    if precision != "planes":
        kk._tile_cache.clear()


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(3, 17, 13, 64), (2, 28, 28, 32), (5, 8, 8, 128)])
def test_batchnorm_relu_maxpool_fused(shape, training):
    """bn_maxpool_fwd / _bwd == BatchNorm apply (+ReLU, sign mask) -> MaxPool2d(3, 2, 1) and their backward chain, the
    stem's bn1 + act1 + maxpool (resnet.py:404-412): pooled values and arg-max taps bit for bit (same fma, same
    first-maximum rule incl. the many exact-zero ties ReLU creates), gradients to fp32 rounding of the fp64 sums."""
    B, H, W, C = shape
    kk = K()
    rng = np.random.default_rng(B * H + C)
    x = torch.from_numpy(rng.standard_normal(shape) * 1.3 - 0.2).float().cuda()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float().cuda()
    beta = torch.from_numpy(rng.standard_normal(C) * 0.3).float().cuda()
    rm = torch.from_numpy(rng.standard_normal(C) * 0.1).float().cuda()
    rv = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float().cuda()
    # unfused chain
    h, saved, mask = kk.bn_fwd(x, gamma, beta, rm.clone(), rv.clone(), training, True, want_mask=True)
    p_ref, arg_ref = kk.maxpool_fwd(h, 3, 2, 1)
    dp = torch.from_numpy(rng.standard_normal(tuple(p_ref.shape))).float().cuda()
    dh = kk.maxpool_bwd(dp, arg_ref, tuple(h.shape), 3, 2, 1)
    dg_ref, db_ref = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx_ref, _ = kk.bn_bwd(dh, None, x, saved, training, dg_ref, db_ref, mask=mask)
    # fused
    saved2 = kk.bn_stats(x, gamma, beta, rm.clone(), rv.clone(), training)
    assert torch.equal(saved2, saved)
    p, arg = kk.bn_maxpool_fwd(x, saved2, 3, 2, 1)
    assert torch.equal(p, p_ref) and torch.equal(arg, arg_ref)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = kk.bn_maxpool_bwd(dp, arg, x, saved2, training, dg, db, 3, 2, 1)
    for a, b in ((dx, dx_ref), (dg, dg_ref), (db, db_ref)):
        sc = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * sc + 1e-7, (float((a - b).abs().max()), sc)


def test_downsample_batchnorm_applied_in_the_last_apply_pass_and_planes_only_outputs():
    """(1) bn_fwd(residual = raw shortcut, residual_bn = its BatchNorm's saved block) == the two apply passes run
    separately, bit for bit (same fma); (2) keep_f32=False writes the same planes and sign mask without the fp32 copy;
    (3) sa_bn_bwd(keep_f32=False) likewise."""
    kk = K()
    rng = np.random.default_rng(5)
    shp = (3, 9, 11, 64)
    rnd = lambda *s_: torch.from_numpy(rng.standard_normal(s_)).float().cuda()
    x, xd = rnd(*shp) * 1.5 + 0.3, rnd(*shp) * 0.7 - 0.1
    C = shp[-1]
    g1, b1, g2, b2 = torch.rand(C, device="cuda") + 0.5, rnd(C), torch.rand(C, device="cuda") + 0.5, rnd(C)
    new = lambda: (torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"))
    y2, saved2 = kk.bn_fwd(xd, g2, b2, *new(), True, False)
    y_ref, saved1, mask_ref = kk.bn_fwd(x, g1, b1, *new(), True, True, residual=y2, want_mask=True)
    saved2b = kk.bn_stats(xd, g2, b2, *new(), True)
    assert torch.equal(saved2b, saved2)
    y, saved1b, mask = kk.bn_fwd(x, g1, b1, *new(), True, True, residual=xd, want_mask=True, residual_bn=saved2b)
    assert torch.equal(y, y_ref) and torch.equal(mask, mask_ref) and torch.equal(saved1b, saved1)
    # planes only
    full, _, m_full = kk.bn_fwd(x, g1, b1, *new(), True, True, want_mask=True, planes=3)
    only, _, m_only = kk.bn_fwd(x, g1, b1, *new(), True, True, want_mask=True, planes=3, keep_f32=False)
    assert only.f32 is None and tuple(only.shape) == shp
    assert torch.equal(only.planes, full.planes) and torch.equal(m_only, m_full)
    rec = only.planes.float().double().sum(0)
    assert torch.equal(rec.float(), full.f32)                      # hi + mid + lo reproduces the fp32 value exactly
    # split-attention backward, planes only
    B, H, W, Cp = 2, 6, 5, 32
    x0 = rnd(B, H, W, 2 * Cp)
    ga, be = torch.rand(2 * Cp, device="cuda") + 0.5, rnd(2 * Cp)
    sv = kk.bn_stats(x0, ga, be, torch.zeros(2 * Cp, device="cuda"), torch.ones(2 * Cp, device="cuda"), True)
    dout, a, dgap = rnd(B, H, W, Cp), torch.rand(B, 2 * Cp, device="cuda"), rnd(B, Cp)
    d_full = kk.sa_bn_bwd(dout, a, dgap, x0, sv, True, planes=3)
    d_only = kk.sa_bn_bwd(dout, a, dgap, x0, sv, True, planes=3, keep_f32=False)
    assert d_only.f32 is None and torch.equal(d_only.planes, d_full.planes)


@pytest.mark.parametrize("case", [(8, 56, 56, 32, 64, 3, 1, 1), (6, 28, 28, 128, 256, 1, 0, 1), (4, 28, 28, 128, 256, 3, 1, 2)])
def test_in_kernel_slab_sum_option(case, monkeypatch):
    """SCOUTER_SLAB_FUSE=1 (off by default: measured slower, kernels.SLAB_FUSE): the last workgroup of
    an output tile sums the split-K slabs inside the weight-gradient kernel.  Same result as the separate slab-sum launch to
    fp32 rounding (another association order), deterministic, arrival counters left zero, scratch poisoned in between."""
    B, H, W, Cin, Cout, k, pad, g = case
    kk = K()
    rng = np.random.default_rng(sum(case))
    x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda()
    ref = torch.empty((k, k, Cin // g, Cout), dtype=torch.float32, device="cuda")
    monkeypatch.setattr(kk, "SLAB_FUSE", False)
    kk.conv2d_wgrad(x, dy, ref, 1, pad, g)
    monkeypatch.setattr(kk, "SLAB_FUSE", True)
    outs = []
    for poison in (float("nan"), 1e30):
        for w in kk._ws.values():
            w.view(torch.float32)[:w.numel() // 4].fill_(poison)
        dw = torch.full_like(ref, float("nan"))
        kk.conv2d_wgrad(x, dy, dw, 1, pad, g)
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    sc = float(ref.abs().max())
    assert float((outs[0] - ref).abs().max()) <= 1e-5 * sc
    for a in kk._arrival.values():
        assert int(a.abs().sum()) == 0
    if (Cin // g) % 64 == 0 and (Cout // g) % 64 == 0:               # the plane kernels, incl. the tap-fused plan
        xp, dyp = kk.planes_split(x, 3), kk.planes_split(dy, 3)
        key = ("pwgrad", 3, B, H, W, Cin, Cout, k, k, pad, g)
        try:
            for plan in (-1, 64) if k == 3 else (-1,):
                kk._tile_cache[key] = plan
                monkeypatch.setattr(kk, "SLAB_FUSE", False)
                r3 = torch.empty_like(ref); kk.conv2d_wgrad_planes(xp, dyp, r3, pad, g)
                monkeypatch.setattr(kk, "SLAB_FUSE", True)
                d3 = torch.full_like(ref, float("nan")); kk.conv2d_wgrad_planes(xp, dyp, d3, pad, g)
                d3b = torch.full_like(ref, float("nan")); kk.conv2d_wgrad_planes(xp, dyp, d3b, pad, g)
                assert torch.equal(d3, d3b)
                assert float((d3 - r3).abs().max()) <= 1e-5 * sc, plan
        finally:
            kk._tile_cache.pop(key, None)


@pytest.mark.parametrize("case", [(3, 20, 20, 64, 256), (2, 13, 11, 64, 64), (5, 9, 9, 128, 512), (2, 17, 15, 256, 64),
                                  (70, 14, 14, 256, 128), (1, 5, 5, 128, 128), (9, 28, 28, 64, 128)])
def test_persistent_pointwise_kernel(case):
    """Tile 4 (csrc/conv_pw_persist.h: weights resident in LDS, A through an LDS-DMA ring, direct stores, statistics
    accumulated per workgroup): forward bit-identical to the implicit-GEMM kernel (same K order per output element), fused
    BatchNorm partial rows sum to the output's fp64 column sums, input gradient (+ addend) bit-identical; ragged M (the last
    tile's rows beyond M must neither be stored nor counted) with a guard band behind the output."""
    B, H, W, Cin, Cout = case
    kk = K()
    rng = np.random.default_rng(sum(case))
    M = B * H * W
    x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)).cuda()
    L = kk._native.lib()
    st = kk._stream()

    def fwd(tile, guard=False):
        rows = L.scouter_conv2d_fwd_bn_partial_rows(B, H, W, Cin, Cout, 1, 1, 1, 0, 1, tile)
        part = torch.full((rows, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
        buf = torch.full((M * Cout + 4096,), 777.0, device="cuda")
        y = buf[:M * Cout].view(B, H, W, Cout)
        kk._native.check(L.scouter_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), part.data_ptr(), B, H, W,
                                                  Cin, Cout, 1, 1, 1, 0, 1, 0, tile, st), "fwd")
        torch.cuda.synchronize()
        assert bool((buf[M * Cout:] == 777.0).all()), "stores beyond the output tensor"
        return y, part
    y2, p2 = fwd(2)
    y4, p4 = fwd(4)
    assert torch.equal(y2, y4)
    yd = y4.double().view(-1, Cout)
    s = p4.sum(0)
    torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-11, atol=1e-9)
    torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-11, atol=1e-9)
    # the wrapper: statistics of tile 4 feed the BatchNorm like any other tile's
    key = ("fwd", False, B, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    try:
        kk._tile_cache[key] = 4
        yw, (pw, rw) = kk.conv2d_fwd(x, w, None, None, 1, 0, 1, False, True)
        assert torch.equal(yw, y4) and rw == p4.shape[0]
        assert torch.equal(kk.conv2d_fwd(x, w, None, None, 1, 0, 1, False, False), y4)     # eval forward: no statistics
        yb = kk.conv2d_fwd(x, w, torch.ones(Cout, device="cuda"), None, 1, 0, 1, True)     # bias + ReLU: not tile 4's business
        assert torch.equal(yb, torch.relu(y4 + 1.0))
    finally:
        kk._tile_cache.pop(key, None)
    # input gradient of the transposed problem: dX [M, Cin] = dY [M, Cout] W^T, with and without the shortcut gradient
    if Cout in (64, 128, 256) and Cin % 64 == 0:
        dy = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda()
        add = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
        for a in (None, add):
            outs = []
            for tile in (2, 4):
                buf = torch.full((M * Cin + 4096,), 777.0, device="cuda")
                dx = buf[:M * Cin].view(B, H, W, Cin)
                kk._native.check(L.scouter_conv2d_dgrad_f32(dy.data_ptr(), w.data_ptr(), None if a is None else a.data_ptr(),
                                                            dx.data_ptr(), B, H, W, Cin, Cout, 1, 1, 1, 0, 1, tile, st), "dgrad")
                torch.cuda.synchronize()
                assert bool((buf[M * Cin:] == 777.0).all())
                outs.append(dx)
            assert torch.equal(outs[0], outs[1]), ("dgrad", a is not None)
    # a request the persistent kernel does not cover is refused, not re-routed
    rc = L.scouter_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), torch.ones(Cout, device="cuda").data_ptr(), None, y4.data_ptr(), None,
                                  B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 0, 4, st)
    assert rc != 0
