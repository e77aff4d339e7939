"""Pointwise convolutions on the register-split bf16x3 GEMM (scouter_amd/csrc/conv_x3.hip): fp32 tensors in and out, the
three-way bf16 split of the activation done in registers, weight planes by LDS-DMA.  The kernel keeps pconv_kernel's K
order, product order and accumulator sets, so it must equal -- BIT FOR BIT -- the plane kernels on the pre-split
activation, for every block tile, incl. ragged last tiles, the fused BatchNorm statistics, addend / ReLU and the fused
BatchNorm-backward epilogue; against an fp64 convolution it is at least as close as the exact-fp32 MFMA kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def K():
    from scouter_amd import kernels
    return kernels


CASES = [  # B, H, W, Cin, Cout
    (2, 14, 14, 256, 1024), (3, 7, 7, 512, 2048), (5, 7, 7, 2048, 512), (1, 28, 28, 128, 512), (2, 28, 27, 512, 128),
    (1, 9, 9, 64, 64), (7, 14, 14, 1024, 256), (1, 1, 37, 64, 128), (33, 28, 27, 256, 512), (1, 3, 5, 96, 192)]


def _tiles(kk, n):
    return [t for t in kk._X3_TILES if kk._x3_tile_ok(t, n)]


@pytest.mark.parametrize("cfg", CASES)
def test_forward_equals_the_plane_kernels_bit_for_bit(cfg):
    B, H, W, Cin, Cout = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg))
    x = torch.from_numpy((rng.standard_normal((B, H, W, Cin)) + 0.2).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)).cuda()
    add = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda()
    wf, wd = kk.planes_split_weight(w, 1, 3)
    xp = kk.planes_split(x, 3)
    ptile = kk._plane_tiles(Cout)[0]
    y_ref, (pr, rr) = kk.conv2d_fwd_planes(xp, wf, 1, 1, 1, 0, 1, bn_stats=True, tile=ptile)
    y_add = kk.conv2d_fwd_planes(xp, wf, 1, 1, 1, 0, 1, addend=add, relu=True, tile=ptile)
    ref64 = (x.double().reshape(-1, Cin) @ w.double().reshape(Cin, Cout)).reshape(B, H, W, Cout)
    y32 = kk.conv2d_fwd(x, w, None, None, 1, 0, 1)
    e32 = float((y32.double() - ref64).abs().max())
    for t in _tiles(kk, Cout):
        y, (part, rows) = kk.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t)
        assert torch.equal(y, y_ref), (cfg, t, float((y - y_ref).abs().max()))
        # statistics: per-tile fp64 partials, grouped by this kernel's M tiles -- equal after the sum to fp64 rounding
        assert part.shape[0] == rows
        st, st_ref = part.sum(0), pr.sum(0)
        assert torch.allclose(st, st_ref, rtol=1e-12, atol=1e-9), (cfg, t)
        ya = kk.conv2d_fwd_x3(x, wf, addend=add, relu=True, tile=t)
        assert torch.equal(ya, y_add), (cfg, t)
    # fp32-grade: no worse than the exact-fp32 MFMA kernel against fp64 (in fact about a third of its error)
    ex = float((y_ref.double() - ref64).abs().max())
    assert ex <= max(e32, 1e-6) * 1.05, (ex, e32)
    # default tile (library heuristic / static table) gives the same bits
    assert torch.equal(kk.conv2d_fwd_x3(x, wf), y_ref)


@pytest.mark.parametrize("cfg", [c for c in CASES if c[3] % 64 == 0])      # (the plane reference needs 64-multiples of Cin)
def test_input_gradient_equals_the_plane_kernels_bit_for_bit(cfg):
    B, H, W, Cin, Cout = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg) + 1)
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)).cuda()
    add = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
    wf, wd = kk.planes_split_weight(w, 1, 3)
    dyp = kk.planes_split(dy, 3)
    ptile = kk._plane_tiles(Cin)[0]
    xs = (B, H, W, Cin)
    dx_ref = kk.conv2d_dgrad_planes(dyp, wd, xs, 1, 1, 1, 0, 1, tile=ptile)
    dx_add = kk.conv2d_dgrad_planes(dyp, wd, xs, 1, 1, 1, 0, 1, addend=add, tile=ptile)
    ref64 = (dy.double().reshape(-1, Cout) @ w.double().reshape(Cin, Cout).t()).reshape(xs)
    d32 = kk.conv2d_dgrad(dy, w, xs, None, 1, 0, 1)
    e32 = float((d32.double() - ref64).abs().max())
    for t in _tiles(kk, Cin):
        dx = kk.conv2d_dgrad_x3(dy, wd, xs, tile=t)
        assert torch.equal(dx, dx_ref), (cfg, t, float((dx - dx_ref).abs().max()))
        assert torch.equal(kk.conv2d_dgrad_x3(dy, wd, xs, addend=add, tile=t), dx_add), (cfg, t)
    ex = float((dx_ref.double() - ref64).abs().max())
    assert ex <= max(e32, 1e-6) * 1.05, (ex, e32)


@pytest.mark.parametrize("two", [False, True])
def test_fused_batchnorm_backward_epilogue(two):
    """The input gradient that produces a block-output gradient masks it with the ReLU sign bits and reduces the
    BatchNorm-backward sums of one or two BatchNorms in its epilogue: same masked gradient bits as the plane kernel's
    fused launch, the same sums (fp64 partials, other tile grouping) -- and both agree with the fp32 kernel's fused launch
    to fp32 rounding."""
    kk = K()
    B, H, W, Cin, Cout = 3, 14, 13, 512, 128
    rng = np.random.default_rng(77 + two)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    dy, w, add = f(B, H, W, Cout), f(1, 1, Cin, Cout) / np.sqrt(Cin), f(B, H, W, Cin)
    xs = (B, H, W, Cin)
    # a real BatchNorm forward provides the saved block and the ReLU mask
    def bn(xin):
        g_, b_ = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        return kk.bn_fwd(xin, g_, b_, rm, rv, True, True, want_mask=True)
    x1, x2 = f(*xs), f(*xs)
    out1 = bn(x1)
    saved1, mask = out1[1], out1[2]
    saved2 = bn(x2)[1]
    entries = [(x1, saved1)] + ([(x2, saved2)] if two else [])
    wf, wd = kk.planes_split_weight(w, 1, 3)
    dyp = kk.planes_split(dy, 3)
    res = {}
    for name in ("fp32", "planes", "x3"):
        post = kk.BnBwdFuse(mask, entries)
        if name == "fp32":
            dx = kk.conv2d_dgrad(dy, w, xs, add, 1, 0, 1, post=post)
        elif name == "planes":
            dx = kk.conv2d_dgrad_planes(dyp, wd, xs, 1, 1, 1, 0, 1, addend=add, post=post, tile=kk._plane_tiles(Cin)[0])
        else:
            dx = kk.conv2d_dgrad_x3(dy, wd, xs, addend=add, post=post)
        assert post.applied
        res[name] = (dx, [p[:post.rows].sum(0) for p in post.parts])
    assert torch.equal(res["x3"][0], res["planes"][0])
    for a, b in zip(res["x3"][1], res["planes"][1]):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-9)
    scale = float(res["fp32"][0].abs().max())
    assert float((res["x3"][0] - res["fp32"][0]).abs().max()) <= 2e-6 * scale
    for a, b in zip(res["x3"][1], res["fp32"][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * float(b.abs().max()))


@pytest.mark.parametrize("case", [(3, 20, 19, 256, 64, True, True), (5, 14, 14, 512, 128, False, True), (2, 28, 28, 256, 128, True, False),
                                  (2, 14, 13, 1024, 256, False, True), (9, 31, 29, 64, 64, True, False), (40, 56, 56, 256, 64, False, True),
                                  (36, 28, 28, 512, 256, True, True)])
def test_persistent_fused_input_gradient_on_the_bf16_matrix_cores(case, monkeypatch):
    """Tile 5 of the fp32 fused input gradient (csrc/conv_pw_persist_x3.h: persistent, dY split three-way in registers, W^T
    planes resident in LDS, v_mfma_f32_16x16x32_bf16 with permuted columns, register epilogue) against the exact-fp32 MFMA
    kernel (tile 2) and an fp64 product: at least as close to fp64 as the fp32 kernel, the same ReLU masking (exact zeros in
    the same places), the same fp64 sums to the rounding of the masked gradient; ragged M, one / two BatchNorms, with /
    without the shortcut gradient, more tiles than workgroups with two workgroups per CU (the last two cases)."""
    B, H, W, Cin, Cout, two, with_add = case
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    rng = np.random.default_rng(sum(case[:5]) + 13)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    shape = (B, H, W, Cin)
    dy, w = f(B, H, W, Cout), f(1, 1, Cin, Cout) / np.sqrt(Cout)
    add = f(*shape) if with_add else None
    g_, b_ = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    xs = [f(*shape) for _ in range(2 if two else 1)]
    saved, mask = [], None
    for i, xh in enumerate(xs):
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        o = kk.bn_fwd(xh, g_, b_, rm, rv, True, True, want_mask=True)
        saved.append(o[1])
        mask = o[2] if i == 0 else mask
    key = ("dgrad+bn", len(xs), with_add, False, B, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    out = {}
    for tile in (2, 5):
        monkeypatch.setitem(kk._tile_cache, key, tile)
        post = kk.BnBwdFuse(mask, list(zip(xs, saved)))
        g = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, post=post)
        assert post.applied
        out[tile] = (g, [p_[:post.rows].sum(0) for p_ in post.parts], post.rows)
    L = kk._native.lib()
    assert out[5][2] == L.scouter_conv2d_dgrad_bn_partial_rows(B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 5) != out[2][2]
    g2, g5 = out[2][0], out[5][0]
    # (an unmasked element may cancel to an exact zero in one of the two roundings: a handful in ten million)
    assert int(((g2 == 0) != (g5 == 0)).sum()) <= 1 + g2.numel() // 1000000 and 0.2 < float((g2 == 0).float().mean()) < 0.8
    ref = dy.double().reshape(-1, Cout) @ w.double().reshape(Cin, Cout).t()
    if with_add:
        ref = ref + add.double().reshape(-1, Cin)
    ref = torch.where((g2.reshape(-1, Cin) == 0) | (g5.reshape(-1, Cin) == 0), torch.zeros_like(ref), ref)
    live = (ref != 0).double()
    e2, e5 = float(((g2.reshape(-1, Cin).double() - ref) * live).abs().max()), float(((g5.reshape(-1, Cin).double() - ref) * live).abs().max())
    assert e5 <= e2 + 1e-7, (e5, e2)
    assert float(((g5.reshape(-1, Cin).double() - ref) * live).pow(2).mean()) <= float(((g2.reshape(-1, Cin).double() - ref) * live).pow(2).mean()) * 1.01
    scale = ref.abs().sum(0)
    for a, b in zip(out[2][1], out[5][1]):
        assert bool(((a[:, 0] - b[:, 0]).abs() <= 1e-6 * scale + 1e-9).all())
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    # no ReLU between the BatchNorm and the consumer (mask None): the unmasked gradient
    nm = {}
    for tile in (2, 5):
        monkeypatch.setitem(kk._tile_cache, key, tile)
        post = kk.BnBwdFuse(None, list(zip(xs, saved)))
        nm[tile] = (kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, post=post), [p_[:post.rows].sum(0) for p_ in post.parts])
    assert float((nm[2][0] - nm[5][0]).abs().max()) <= 4e-6 * float(nm[2][0].abs().max())
    for a, b in zip(nm[2][1], nm[5][1]):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    # named where it does not apply: an error, not a re-route
    rc = L.scouter_conv2d_dgrad_bnbwd_f32(kk._p(dy), kk._p(w), None, kk._p(torch.empty(shape, device="cuda")), B, H, W, Cin, Cout, 1, 1,
                                          1, 0, 1, 5, *kk._NO_FUSE, None)
    assert rc != 0 and "tile 5" in L.scouter_last_error().decode()


def test_model_routes_the_deep_pointwise_layers_and_the_switch_turns_them_off():
    """SlotModel.set_x3: the 14 deep 1x1 convolutions of resnest26d (Cin * Cout >= 2^16) run on the register-split
    GEMM -- a static rule of the layer shapes -- and with the switch off every one is back on the fp32 MFMA kernels;
    the two forwards agree to fp32 rounding (both are fp32-grade products), neither depends on the batch."""
    import test_model_gpu as T
    from scouter_amd import _native
    res = {}
    for bits in (63, 31, 15, 0):
        m, P, images, labels, cfg = T._synthetic_model("resnest26d", 10, 1, 3, 4, 96, 2500)
        m.set_x3(bits)
        assert len(m._x3_convs) == ((15 if bits & 16 else 14) if bits else 0)       # 14 deep 1x1 layers + the stem's 32 -> 64 3x3
        # bit 5: the forward of the eight short-K pointwise layers (layer1's five, layer2's 256 -> 128 and its two 128 -> 512)
        assert sum(c.fwd_on_xpw() for c in m.backbone.modules() if hasattr(c, "fwd_on_xpw")) == (8 if bits & 32 else 0)
        L = _native.lib()
        L.scouter_prof_enable(1)
        out, (loss, nll, area) = m(images.cuda(), labels.cuda())
        loss.backward()
        torch.cuda.synchronize()
        L.scouter_prof_enable(0)
        import ctypes
        buf = ctypes.create_string_buffer(1 << 16)
        L.scouter_prof_collect(buf, len(buf))
        names = buf.value.decode()
        assert ("xconv_fwd<bf16x3>" in names) == bool(bits) and ("xconv_dgrad<bf16x3>" in names) == bool(bits)
        assert ("xwgrad<bf16x3>" in names) == bool(bits)
        assert ("xpw_fwd<bf16x3>" in names) == bool(bits & 32)
        res[bits] = (out.detach().clone(), m.grad_arena().flat.clone())
    assert float((res[31][0] - res[0][0]).abs().max()) < 5e-4 and float((res[15][0] - res[0][0]).abs().max()) < 5e-4
    assert float((res[63][0] - res[0][0]).abs().max()) < 5e-4
    rel = float((res[31][1] - res[0][1]).norm() / res[0][1].norm())
    assert rel < 5e-2, rel          # (random-init net: ReLU sign flips on ~0 pre-activations move gradients, see test_model_gpu)


WCASES = [  # B, H, W, Cin, Cout
    (70, 7, 7, 512, 2048), (8, 14, 14, 1024, 256), (3, 28, 28, 128, 512), (2, 9, 9, 256, 128), (1, 5, 5, 128, 128), (5, 17, 17, 256, 1024)]


@pytest.mark.parametrize("cfg", WCASES)
def test_weight_gradient_matches_fp64_at_least_as_well_as_the_fp32_kernel(cfg):
    """dW = X^T dY over the pixels with both operands transposed and split three-way in registers: against an fp64 product it
    is at least as close as the exact-fp32 MFMA weight gradient, for every split-K plan (incl. pixel counts that are no
    multiple of the 32-pixel chunk and splits whose last chunk is ragged), and bit-reproducible launch to launch on a
    poisoned workspace."""
    B, H, W, Cin, Cout = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg) + 5)
    x = torch.from_numpy((rng.standard_normal((B, H, W, Cin)) + 0.1).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda()
    ref = x.double().reshape(-1, Cin).t() @ dy.double().reshape(-1, Cout)
    dw32 = torch.empty(1, 1, Cin, Cout, device="cuda")
    kk.conv2d_wgrad(x, dy, dw32, 1, 0, 1)
    e32 = float((dw32.double().view(Cin, Cout) - ref).abs().max())
    for plan in kk._X3_WGRAD_PLANS:
        dw = torch.full((1, 1, Cin, Cout), float("nan"), device="cuda")
        kk.conv2d_wgrad_x3(x, dy, dw, plan=plan)
        ex = float((dw.double().view(Cin, Cout) - ref).abs().max())
        assert ex <= max(e32, 1e-6) * 1.05, (cfg, plan, ex, e32)
        ws = kk.workspace(1, x.device)
        ws.fill_(0xFF)                                   # NaN patterns: nothing may be read before it is written
        dw2 = torch.full((1, 1, Cin, Cout), float("nan"), device="cuda")
        kk.conv2d_wgrad_x3(x, dy, dw2, plan=plan)
        assert torch.equal(dw, dw2), (cfg, plan)


CCASES = [  # B, H, W, Cin, Cout, groups   (3x3 / stride 1 / pad 1)
    (2, 20, 20, 32, 64, 1), (3, 12, 9, 64, 128, 2), (1, 33, 31, 32, 32, 1), (2, 7, 7, 64, 64, 2), (5, 16, 16, 32, 64, 1),
    (1, 56, 56, 64, 128, 2), (2, 9, 12, 128, 256, 2)]


@pytest.mark.parametrize("cfg", CCASES)
def test_3x3_layers_forward_and_input_gradient(cfg):
    """The same kernel as an implicit GEMM over filter taps (padding rows read zeros through out-of-range offsets, groups,
    32-column tiles): the forward equals the plane kernels bit for bit where those support the shape (64-multiples of output
    channels per group); forward and input gradient are at least as close to an fp64 convolution as the exact-fp32 kernel;
    every tile gives the same bits; the fused BatchNorm-backward epilogue agrees with the fp32 kernel's."""
    import torch.nn.functional as F
    B, H, W, Cin, Cout, G = cfg
    kk = K()
    rng = np.random.default_rng(sum(cfg) + 9)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    x, dy = f(B, H, W, Cin) + 0.2, f(B, H, W, Cout)
    w = f(3, 3, Cin // G, Cout) / np.sqrt(9 * Cin // G)                      # HWIO
    wf, wd = kk.planes_split_weight(w, G, 3)
    w_oihw = w.permute(3, 2, 0, 1).double().cpu()
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, w_oihw, None, 1, 1, 1, G)
    yr.backward(dy.permute(0, 3, 1, 2).double().cpu())
    y_ref, dx_ref = yr.detach().permute(0, 2, 3, 1).cuda(), xr.grad.permute(0, 2, 3, 1).cuda()
    y32 = kk.conv2d_fwd(x, w, None, None, 1, 1, G)
    d32 = kk.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 1, G)
    e32y, e32d = float((y32.double() - y_ref).abs().max()), float((d32.double() - dx_ref).abs().max())
    ys, ds = [], []
    for t in _tiles(kk, Cout // G):
        yt, (part, rows) = kk.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t, kh=3, pad=1, groups=G)
        ys.append(yt)
        st = part[:rows].sum(0)
        assert torch.allclose(st[:, 0], yt.double().reshape(-1, Cout).sum(0), rtol=1e-9, atol=1e-6)
    for t in _tiles(kk, Cin // G):
        ds.append(kk.conv2d_dgrad_x3(dy, wd, tuple(x.shape), tile=t, kh=3, pad=1, groups=G))
    for yt in ys[1:]:
        assert torch.equal(yt, ys[0])
    for dt in ds[1:]:
        assert torch.equal(dt, ds[0])
    assert float((ys[0].double() - y_ref).abs().max()) <= max(e32y, 1e-6) * 1.05
    assert float((ds[0].double() - dx_ref).abs().max()) <= max(e32d, 1e-6) * 1.05
    if (Cout // G) % 64 == 0:
        xp = kk.planes_split(x, 3)
        yp = kk.conv2d_fwd_planes(xp, wf, 3, 3, 1, 1, G, tile=kk._plane_tiles(Cout // G)[0])
        assert torch.equal(ys[0], yp)
    # fused BatchNorm-backward epilogue against the fp32 kernel's
    C = Cin
    g_, b_ = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    _, saved, mask = kk.bn_fwd(x, g_, b_, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True, True, want_mask=True)
    res = []
    for name in ("fp32", "x3"):
        post = kk.BnBwdFuse(mask, [(x, saved)])
        if name == "fp32":
            dxf = kk.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 1, G, post=post)
        else:
            dxf = kk.conv2d_dgrad_x3(dy, wd, tuple(x.shape), post=post, kh=3, pad=1, groups=G)
        assert post.applied
        res.append((dxf, post.parts[0][:post.rows].sum(0)))
    scale = float(res[0][0].abs().max())
    assert float((res[1][0] - res[0][0]).abs().max()) <= 3e-6 * scale
    assert torch.allclose(res[1][1], res[0][1], rtol=1e-4, atol=1e-3 * float(res[0][1].abs().max()))
