"""bf16 ACTIVATION STORAGE of the bf16 mode (-m gpu): the typed (`*_io`) entry points of include/scouter_hip.h.

Storage is not arithmetic: a kernel that reads a bf16-stored tensor must give, BIT FOR BIT, what the fp32-storage kernel
gives on the widened tensor, and a kernel that stores bf16 must store the RNE rounding of what the fp32-storage kernel
writes.  Every test below is that statement for one entry point (torch's .float() / .to(bfloat16) are the widening and
the RNE rounding).  The model-level effect is covered by tests/test_model_gpu.py (oracle ACTIVATION_STORAGE emulation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def K():
    from scouter_amd import kernels
    return kernels


@pytest.fixture(autouse=True)
def _every_layer_on_the_bf16_kernels(monkeypatch):
    monkeypatch.setattr(K(), "BF16_MIN_PIXELS", 1)       # (the model keeps layers under 1024 pixels on the fp32 kernels)


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape) * scale).float().cuda()


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, pad, groups
    (3, 20, 19, 256, 64, 1, 0, 1),      # conv1 of a bottleneck: bf16-stored block input (K = 256)
    (5, 17, 17, 64, 256, 1, 0, 1),      # conv3: bf16-stored output, 128-wide tiles
    (2, 28, 28, 128, 512, 1, 0, 1),
    (4, 23, 9, 96, 64, 1, 0, 1),        # K = 96: the 32-channel K tile
    (2, 24, 24, 64, 128, 3, 1, 2)])     # a 3x3 layer through the same kernel
def test_conv_forward_typed_storage(case, monkeypatch):
    B, H, W, Cin, Cout, k, pad, groups = case
    kk = K()
    monkeypatch.setattr(kk, "PWB_FWD", False)          # (the tile kernels; the persistent typed forward has its own test below)
    rng = np.random.default_rng(sum(case))
    xh = _rand(rng, B, H, W, Cin).to(BF16)
    w = _rand(rng, k, k, Cin // groups, Cout, scale=0.1)
    key = ("fwd", True, B, H, W, Cin, Cout, k, k, 1, pad, groups)
    for tile in (0, 1, 2, 3):
        if not kk._tile_legal(Cout // groups, tile):
            continue
        kk._tile_cache[key] = tile
        y32, (p32, rows) = kk.conv2d_fwd(xh.float(), w, None, None, 1, pad, groups, False, bn_stats=True, precision="bf16")
        yx, (px, _) = kk.conv2d_fwd(xh, w, None, None, 1, pad, groups, False, bn_stats=True, precision="bf16")
        assert torch.equal(yx, y32) and torch.equal(px, p32), "bf16-stored input: tile %d" % tile
        yo, (po, _) = kk.conv2d_fwd(xh, w, None, None, 1, pad, groups, False, bn_stats=True, precision="bf16",
                                    out_dtype=BF16)
        assert yo.dtype == BF16 and torch.equal(yo, y32.to(BF16)), "bf16-stored output: tile %d" % tile
        assert torch.equal(po, p32), "the statistics are those of the fp32 accumulators"
    del kk._tile_cache[key]
    with pytest.raises(RuntimeError, match="bf16-stored activations"):
        kk.conv2d_fwd(xh, w, None, None, 1, pad, groups, precision="fp32")


@pytest.mark.parametrize("case", [(5, 17, 17, 64, 256), (2, 28, 28, 128, 512), (3, 20, 19, 256, 128), (2, 14, 13, 512, 256),
                                  (40, 56, 56, 64, 256), (36, 28, 28, 256, 512), (9, 28, 27, 128, 128)])
def test_persistent_typed_forward_against_the_tile_kernel(case, monkeypatch):
    """Tile 4 of the typed forward (csrc/conv_pw_persist_bf16.h pwb_fwd_kernel: persistent, v_mfma_f32_16x16x32_bf16 with
    permuted columns, register epilogue) -- the static choice for pointwise layers whose input is stored as bf16 -- against
    the tile kernel: the same bf16 products summed in another order inside the matrix unit: fp32 outputs equal to fp32
    rounding, bf16-stored outputs to one bf16 ulp on a few elements, the fp64 statistics of the fp32 accumulators to that;
    ragged M, more tiles than workgroups with two / three workgroups per CU."""
    B, H, W, Cin, Cout = case
    kk = K()
    rng = np.random.default_rng(sum(case) + 5)
    xh = _rand(rng, B, H, W, Cin).to(BF16)
    w = _rand(rng, 1, 1, Cin, Cout, scale=0.1)
    res = {}
    for on in (False, True):
        monkeypatch.setattr(kk, "PWB_FWD", on)
        y32, (p32, r32) = kk.conv2d_fwd(xh, w, bn_stats=True, precision="bf16")
        y16, (p16, r16) = kk.conv2d_fwd(xh, w, bn_stats=True, precision="bf16", out_dtype=BF16)
        yn = kk.conv2d_fwd(xh, w, precision="bf16", out_dtype=BF16)
        assert torch.equal(yn, y16) and r16 == r32 and torch.equal(p16, p32)
        res[on] = (y32, y16.float(), p32[:r32].sum(0), r32)
    L = kk._native.lib()
    assert res[True][3] == L.scouter_conv2d_fwd_bn_partial_rows_bf16(B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 4) != res[False][3]
    a, b = res[False][0], res[True][0]
    scale = float(a.abs().max())
    assert scale > 0 and float((a - b).abs().max()) <= 2e-6 * scale
    d = (res[False][1] - res[True][1]).abs()
    assert bool((d <= res[False][1].abs() * 2.0 ** -7 + 1e-30).all()) and float((d > 0).float().mean()) < 0.02
    assert torch.equal(res[True][1], b.to(BF16).float()), "the bf16-stored output is the RNE of the fp32 one"
    sa, sb = res[False][2], res[True][2]
    assert float((sa - sb).abs().max()) <= 1e-6 * float(sa.abs().max())
    # named where it does not apply (fp32-stored input): an error, not a re-route
    y = torch.empty(B, H, W, Cout, device="cuda")
    wt = torch.empty(1, Cout, Cin, dtype=BF16, device="cuda")
    rc = L.scouter_conv2d_fwd_bf16_io(kk._p(xh.float()), kk._p(wt), None, None, kk._p(y), None, B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 0, 4, 0, None)
    assert rc != 0 and "tile 4" in L.scouter_last_error().decode()


def _stats_of(x):
    """[1, C, 2] fp64 partial sums (sum, sum of squares) as a convolution epilogue hands them over"""
    xd = x.double().reshape(-1, x.shape[-1])
    return torch.stack([xd.sum(0), (xd * xd).sum(0)], dim=1).unsqueeze(0).contiguous(), 1


@pytest.mark.parametrize("shape", [(3, 14, 14, 256), (2, 9, 7, 64), (5, 28, 28, 512)])
@pytest.mark.parametrize("with_res", ["none", "plain", "downsample"])
def test_batchnorm_apply_typed_storage(shape, with_res):
    kk = K()
    rng = np.random.default_rng(shape[0] + shape[3] + len(with_res))
    C = shape[-1]
    x = _rand(rng, *shape)                      # the convolution result
    xh = x.to(BF16)                             # ... as stored
    gamma, beta = _rand(rng, C).abs() + 0.5, _rand(rng, C)
    res = _rand(rng, *shape).to(BF16) if with_res != "none" else None
    rbn = None
    if with_res == "downsample":                # raw downsample-convolution output + the saved block of its BatchNorm
        rbn = kk.bn_stats(res.float(), _rand(rng, C).abs() + 0.5, _rand(rng, C), torch.zeros(C, device="cuda"),
                          torch.ones(C, device="cuda"), True)
    stats = _stats_of(x)                        # statistics of the UNROUNDED result

    def run(xin, rin, out_dtype):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y, saved, mask = kk.bn_fwd(xin, gamma, beta, rm, rv, True, True, rin, stats=stats, want_mask=True,
                                   residual_bn=rbn, out_dtype=out_dtype)
        return y, saved, mask, rm, rv
    ref = run(xh.float(), None if res is None else res.float(), torch.float32)
    got = run(xh, res, torch.float32)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    got16 = run(xh, res, BF16)
    assert got16[0].dtype == BF16 and torch.equal(got16[0], ref[0].to(BF16))
    assert torch.equal(got16[2], ref[2]), "the ReLU sign bits come from the unrounded value"
    # eval mode (running statistics): no producer statistics needed
    rm, rv = _rand(rng, C) * 0.1, _rand(rng, C).abs() + 0.5
    ye = kk.bn_fwd(xh, gamma, beta, rm, rv, False, True, res, out_dtype=BF16)[0]
    ye_ref = kk.bn_fwd(xh.float(), gamma, beta, rm, rv, False, True, None if res is None else res.float())[0]
    assert torch.equal(ye, ye_ref.to(BF16))
    with pytest.raises(RuntimeError, match="statistics of its producer"):
        kk.bn_fwd(xh, gamma, beta, rm, rv, True, True)


@pytest.mark.parametrize("shape", [(3, 14, 14, 256), (2, 9, 7, 64), (6, 28, 28, 512), (2, 31, 5, 128)])
def test_batchnorm_backward_typed_storage(shape):
    kk = K()
    rng = np.random.default_rng(shape[0] * 7 + shape[3])
    C = shape[-1]
    xh = _rand(rng, *shape).to(BF16)
    gamma, beta = _rand(rng, C).abs() + 0.5, _rand(rng, C)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    _, saved, mask = kk.bn_fwd(xh, gamma, beta, rm, rv, True, True, stats=_stats_of(xh.float()), want_mask=True)
    dy = _rand(rng, *shape)
    outs = []
    for xin in (xh.float(), xh):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dx, gout = kk.bn_bwd(dy, None, xin, saved, True, dg, db, want_gout=True, mask=mask)     # own reduction pass
        outs.append((dx, gout, dg, db))
    for a, b in zip(*outs):
        assert a.dtype == torch.float32 and torch.equal(a, b)
    # producer-reduced sums (ext): finalize + apply only
    g = outs[0][1]
    xhat = (xh.double() - saved[0].double()) * saved[1].double()
    part = torch.stack([g.double().reshape(-1, C).sum(0), (g.double() * xhat).reshape(-1, C).sum(0)], 1).unsqueeze(0).contiguous()
    ext = []
    for xin in (xh.float(), xh):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ext.append(kk.bn_bwd(g, None, xin, saved, True, dg, db, ext=(part, 1))[0])
    assert torch.equal(ext[0], ext[1])
    np.testing.assert_allclose(ext[0].cpu().numpy(), outs[0][0].cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_avgpool_typed_storage():
    kk = K()
    rng = np.random.default_rng(5)
    for shape, (k, s, p, ceil, cip) in (((3, 28, 28, 256), (2, 2, 0, True, False)), ((2, 15, 13, 64), (2, 2, 0, True, False)),
                                        ((2, 14, 14, 128), (3, 2, 1, False, True))):
        xh = _rand(rng, *shape).to(BF16)
        assert torch.equal(kk.avgpool_fwd(xh, k, s, p, ceil, cip), kk.avgpool_fwd(xh.float(), k, s, p, ceil, cip))


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, pad, groups
    (3, 20, 19, 256, 64, 1, 0, 1), (2, 28, 28, 512, 128, 1, 0, 1), (4, 23, 9, 96, 64, 1, 0, 1), (2, 24, 24, 64, 128, 3, 1, 2),
    (2, 30, 30, 128, 128, 3, 1, 1)])
def test_weight_gradient_typed_storage(case):
    B, H, W, Cin, Cout, k, pad, groups = case
    kk = K()
    rng = np.random.default_rng(sum(case) + 1)
    xh = _rand(rng, B, H, W, Cin).to(BF16)
    dy = _rand(rng, B, H, W, Cout)
    key = ("wgrad", True, B, H, W, Cin, Cout, k, k, 1, pad, groups)
    for plan in (-1, 0, 17, 34, 51):
        kk._tile_cache[key] = plan
        dw32 = torch.zeros(k, k, Cin // groups, Cout, device="cuda")
        dwh = torch.zeros_like(dw32)
        kk.conv2d_wgrad(xh.float(), dy, dw32, 1, pad, groups, precision="bf16")
        kk.conv2d_wgrad(xh, dy, dwh, 1, pad, groups, precision="bf16")
        assert torch.equal(dw32, dwh), "plan %d" % plan
    del kk._tile_cache[key]
    with pytest.raises(RuntimeError, match="bf16-stored activation"):
        kk.conv2d_wgrad(xh, dy, dwh, 1, pad, groups, precision="fp32")


@pytest.mark.parametrize("case", [
    # B, H, W, Cin (= channels of the block output whose gradient is produced), Cout, two BatchNorms?, shortcut addend?
    (3, 20, 19, 256, 64, True, True), (5, 14, 14, 512, 128, False, True), (2, 28, 28, 256, 128, True, False),
    (7, 13, 11, 128, 32, False, False)])
def test_fused_batchnorm_backward_epilogue_reads_typed_storage(case, monkeypatch):
    """conv1's input-gradient kernel finishing the previous block's bn3 (+ downsample BatchNorm) sums from bf16-stored
    BatchNorm inputs == the same launch on the widened inputs: masked gradient and fp64 partial sums bit for bit."""
    B, H, W, Cin, Cout, two, with_add = case
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    rng = np.random.default_rng(sum(case[:5]))
    shape = (B, H, W, Cin)
    dy = _rand(rng, B, H, W, Cout)
    w = _rand(rng, 1, 1, Cin, Cout, scale=0.1)
    add = _rand(rng, *shape) if with_add else None
    gamma, beta = _rand(rng, Cin).abs() + 0.5, _rand(rng, Cin)
    xs = [_rand(rng, *shape).to(BF16) for _ in range(2 if two else 1)]
    ctxs = []
    for i, xh in enumerate(xs):
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        if i == 0:
            _, saved, mask = kk.bn_fwd(xh, gamma, beta, rm, rv, True, True, stats=_stats_of(xh.float()), want_mask=True)
        else:
            saved = kk.bn_stats(xh, gamma, beta, rm, rv, True, stats=_stats_of(xh.float()))
        ctxs.append(saved)
    res = []
    for widen in (True, False):
        post = kk.BnBwdFuse(mask, [((x.float() if widen else x), s) for x, s in zip(xs, ctxs)])
        dx = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16", post=post)
        assert post.applied
        res.append((dx, post.parts, post.rows))
    assert torch.equal(res[0][0], res[1][0]) and res[0][2] == res[1][2]
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    # the fp32-input kernel's epilogue does not read bf16 storage: the fuse is left to the BatchNorm's own (typed) backward
    post = kk.BnBwdFuse(mask, [(x, s) for x, s in zip(xs, ctxs)])
    kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="fp32", post=post)
    assert not post.applied


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, pad, groups
    (3, 20, 19, 64, 256, 1, 0, 1), (2, 28, 28, 128, 512, 1, 0, 1), (4, 23, 9, 96, 64, 1, 0, 1), (2, 24, 24, 64, 128, 3, 1, 2)])
def test_gradients_read_only_by_bf16_kernels_may_be_stored_as_bf16(case, monkeypatch):
    """dc = BatchNorm-backward output in front of a convolution whose input- and weight-gradient kernels are the bf16-input
    ones: stored as bf16 it is exactly what those kernels round an fp32 dc to -- dx and dW bit for bit the same."""
    B, H, W, Cin, Cout, k, pad, groups = case
    kk = K()
    monkeypatch.setattr(kk, "PWB_DGRAD", False)        # (the tile kernels; the persistent plain input gradient has its own test)
    rng = np.random.default_rng(sum(case) + 2)
    C = Cout
    x = _rand(rng, B, H, W, Cin)
    w = _rand(rng, k, k, Cin // groups, Cout, scale=0.1)
    y = _rand(rng, B, H, W, C)                          # the convolution output = BatchNorm input
    gamma, beta = _rand(rng, C).abs() + 0.5, _rand(rng, C)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    _, saved, mask = kk.bn_fwd(y, gamma, beta, rm, rv, True, True, want_mask=True)
    dout = _rand(rng, B, H, W, C)
    dc32, _ = kk.bn_bwd(dout, None, y, saved, True, None, None, mask=mask)
    dc16, _ = kk.bn_bwd(dout, None, y, saved, True, None, None, mask=mask, dx_dtype=BF16)
    assert dc16.dtype == BF16 and torch.equal(dc16, dc32.to(BF16))
    dx32 = kk.conv2d_dgrad(dc32, w, tuple(x.shape), None, 1, pad, groups, precision="bf16")
    dx16 = kk.conv2d_dgrad(dc16, w, tuple(x.shape), None, 1, pad, groups, precision="bf16")
    assert torch.equal(dx32, dx16)
    dw32, dw16 = torch.zeros_like(w), torch.zeros_like(w)
    kk.conv2d_wgrad(x, dc32, dw32, 1, pad, groups, precision="bf16")
    kk.conv2d_wgrad(x, dc16, dw16, 1, pad, groups, precision="bf16")
    assert torch.equal(dw32, dw16)
    kk.conv2d_wgrad(x.to(BF16), dc16, dw16, 1, pad, groups, precision="bf16")           # both operands bf16-stored
    kk.conv2d_wgrad(x.to(BF16).float(), dc32, dw32, 1, pad, groups, precision="bf16")
    assert torch.equal(dw32, dw16)
    with pytest.raises(RuntimeError, match="bf16-stored gradients"):
        kk.conv2d_dgrad(dc16, w, tuple(x.shape), None, 1, pad, groups, precision="fp32")


@pytest.mark.parametrize("case", [(3, 20, 19, 64, 256, False), (2, 28, 28, 128, 512, False), (5, 14, 13, 256, 1024, False), (9, 31, 29, 64, 64, True),
                                  (40, 56, 56, 64, 256, False), (36, 28, 28, 256, 512, True), (2, 14, 14, 512, 1024, False)])
def test_persistent_plain_typed_input_gradient_against_the_tile_kernel(case, monkeypatch):
    """Tile 4 of the typed input gradient WITHOUT the fused epilogue (csrc/conv_pw_persist_bf16.h pwb_dgrad_kernel: dy stored
    as bf16, dx fp32, optional addend) -- the static choice for the plain pointwise input gradients of the bf16 mode -- against
    the tile kernel: the same bf16 products summed in another order inside the matrix unit: equal to fp32 rounding; ragged M,
    with / without the addend (fp32 and bf16-stored), K = Cout up to 1024, more tiles than workgroups."""
    B, H, W, Cin, Cout, with_add = case
    kk = K()
    rng = np.random.default_rng(sum(case[:5]) + 17)
    dy = _rand(rng, B, H, W, Cout).to(BF16)
    w = _rand(rng, 1, 1, Cin, Cout, scale=0.1)
    shape = (B, H, W, Cin)
    adds = [None] if not with_add else [_rand(rng, *shape), _rand(rng, *shape).to(BF16)]
    for add in adds:
        res = {}
        for on in (False, True):
            monkeypatch.setattr(kk, "PWB_DGRAD", on)
            res[on] = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16")
            assert res[on].dtype == torch.float32
        scale = float(res[False].abs().max())
        assert scale > 0 and float((res[False] - res[True]).abs().max()) <= 2e-6 * scale
    # named where it does not apply (fp32-stored dy): an error, not a re-route
    L = kk._native.lib()
    rc = L.scouter_conv2d_dgrad_bnbwd_bf16_io(kk._p(dy.float()), kk._p(w), None, kk._p(torch.empty(shape, device="cuda")), B, H, W, Cin,
                                              Cout, 1, 1, 1, 0, 1, 4, *kk._NO_FUSE, 0, None)
    assert rc != 0 and "tile 4" in L.scouter_last_error().decode()


def test_gradient_storage_changes_no_bit_of_the_model(monkeypatch):
    """Whole model (resnest26d + xSlot, precision bf16, batch 8 x 224 x 224 so that layer1-3 run the bf16 kernels): with the
    gradients in front of conv1 / conv3 / the downsample convolutions stored as bf16 (nn_hip.Conv2d.grad_storage) every
    output and every parameter gradient is the same bits as with fp32 storage."""
    import test_model_gpu as T
    from scouter_amd import nn_hip
    monkeypatch.undo()                                   # (the model's own pixel rule, not the kernel tests' override)
    res = []
    for flag in (False, True):
        monkeypatch.setattr(nn_hip, "GRAD_STORAGE_BF16", flag)
        m, P, images, labels, cfg = T._synthetic_model("resnest26d", 10, 1, 3, 8, 224, 1900)
        m.set_precision("bf16")
        out, (loss, nll, area) = m(images.cuda(), labels.cuda())
        loss.backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), m.grad_arena().flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize("case", [(3, 20, 19, 256, 64, True, True), (5, 14, 14, 512, 128, False, True), (2, 28, 28, 256, 128, True, False),
                                  (2, 14, 13, 1024, 256, False, True), (3, 7, 7, 2048, 512, True, True), (9, 31, 29, 128, 64, False, False),
                                  (40, 56, 56, 256, 64, False, True)])
def test_persistent_typed_fused_input_gradient_against_the_tile_kernel(case, monkeypatch):
    """Tile 4 of the typed input gradient (csrc/conv_pw_persist_bf16.h: persistent, v_mfma_f32_16x16x32_bf16 with permuted
    columns, register epilogue) against tile 1 (igemm_bf16_kernel): the same products summed over K in another order inside
    the matrix unit -- g equal to an fp32 ulp before its RNE to bf16, i.e. to ONE bf16 ulp on a few elements; the fp64 sums
    (one partial row per workgroup instead of one per M tile) equal to that.  Ragged M, one / two BatchNorms, with / without
    the shortcut gradient, more tiles than workgroups (last case)."""
    B, H, W, Cin, Cout, two, with_add = case
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    rng = np.random.default_rng(sum(case[:5]) + 11)
    shape = (B, H, W, Cin)
    dy = _rand(rng, B, H, W, Cout).to(BF16)
    w = _rand(rng, 1, 1, Cin, Cout, scale=0.1)
    add = _rand(rng, *shape).to(BF16) if with_add else None
    gamma, beta = _rand(rng, Cin).abs() + 0.5, _rand(rng, Cin)
    xs = [_rand(rng, *shape).to(BF16) for _ in range(2 if two else 1)]
    saved = []
    for i, xh in enumerate(xs):
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        if i == 0:
            _, sv, mask = kk.bn_fwd(xh, gamma, beta, rm, rv, True, True, stats=_stats_of(xh.float()), want_mask=True)
        else:
            sv = kk.bn_stats(xh, gamma, beta, rm, rv, True, stats=_stats_of(xh.float()))
        saved.append(sv)
    key = ("dgrad+bn", len(xs), with_add, True, B, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    out = {}
    for tile in (1, 4):
        monkeypatch.setitem(kk._tile_cache, key, tile)
        post = kk.BnBwdFuse(mask, list(zip(xs, saved)))
        g = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16", post=post, out_dtype=BF16)
        assert post.applied and g.dtype == BF16
        out[tile] = (g.float(), [p_[:post.rows].sum(0) for p_ in post.parts], post.rows)
    L = kk._native.lib()
    assert out[4][2] == L.scouter_conv2d_dgrad_bn_partial_rows_bf16(B, H, W, Cin, Cout, 1, 1, 1, 0, 1, 4, int(two))
    assert out[4][2] != out[1][2]
    g1, g4 = out[1][0], out[4][0]
    assert float(g1.abs().max()) > 0
    d = (g1 - g4).abs()
    # (2e-6: where the product and the shortcut gradient cancel, an fp32 ulp of the product is not small against g)
    assert bool((d <= g1.abs() * 2.0 ** -7 + 2e-6).all()), "more than one bf16 ulp"
    assert float((d > 0).float().mean()) < 0.02, "an fp32 ulp before the rounding flips a bf16 ulp on a few elements only"
    scale = g1.abs().double().sum((0, 1, 2))
    for a, b in zip(out[1][1], out[4][1]):
        assert bool(((a[:, 0] - b[:, 0]).abs() <= 2.0 ** -8 * 0.02 * scale + 1e-9).all())
        assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max())
    # no ReLU between the BatchNorm and the consumer (mask None): the unmasked gradient
    nm = {}
    for tile in (1, 4):
        monkeypatch.setitem(kk._tile_cache, key, tile)
        post = kk.BnBwdFuse(None, list(zip(xs, saved)))
        nm[tile] = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16", post=post, out_dtype=BF16).float()
    assert bool(((nm[1] - nm[4]).abs() <= nm[1].abs() * 2.0 ** -7 + 2e-6).all()) and float(nm[1].abs().max()) > 0
    # anything the persistent kernel does not cover, named with tile 4, is an error -- never a silent re-route
    post = kk.BnBwdFuse(mask, [(xs[0].float(), saved[0])])
    rc = L.scouter_conv2d_dgrad_bnbwd_bf16_io(kk._p(dy), kk._p(w), None, kk._p(torch.empty(shape, dtype=BF16, device="cuda")), B, H, W,
                                              Cin, Cout, 1, 1, 1, 0, 1, 4, *kk._NO_FUSE, kk.DGRAD_IO_DY | kk.DGRAD_IO_DX, None)
    assert rc != 0 and "tile 4" in L.scouter_last_error().decode()


@pytest.mark.parametrize("case", [(3, 20, 19, 256, 64, True, True), (5, 14, 14, 512, 128, False, True), (2, 28, 28, 256, 128, True, False)])
def test_masked_block_gradient_stored_as_bf16(case, monkeypatch):
    """The residual-stream gradient: conv1's fused input-gradient epilogue stores the masked gradient g as bf16 (= RNE of the
    fp32-storage launch's g, same partial sums), takes a bf16-stored shortcut addend, and the BatchNorm backward / the
    next epilogue read the stored g like the widened tensor."""
    B, H, W, Cin, Cout, two, with_add = case
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    rng = np.random.default_rng(sum(case[:5]) + 3)
    shape = (B, H, W, Cin)
    dy = _rand(rng, B, H, W, Cout)
    w = _rand(rng, 1, 1, Cin, Cout, scale=0.1)
    add = _rand(rng, *shape).to(BF16) if with_add else None
    gamma, beta = _rand(rng, Cin).abs() + 0.5, _rand(rng, Cin)
    xs = [_rand(rng, *shape).to(BF16) for _ in range(2 if two else 1)]
    saved = []
    for i, xh in enumerate(xs):
        rm, rv = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda")
        if i == 0:
            _, sv, mask = kk.bn_fwd(xh, gamma, beta, rm, rv, True, True, stats=_stats_of(xh.float()), want_mask=True)
        else:
            sv = kk.bn_stats(xh, gamma, beta, rm, rv, True, stats=_stats_of(xh.float()))
        saved.append(sv)
    out = []
    for typed in (False, True):
        post = kk.BnBwdFuse(mask, list(zip(xs, saved)))
        g = kk.conv2d_dgrad(dy, w, shape, (add if typed else add.float()) if with_add else None, 1, 0, 1, precision="bf16",
                            post=post, out_dtype=BF16 if typed else torch.float32)
        assert post.applied and g.dtype == (BF16 if typed else torch.float32)
        dg, db = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        gin = g if typed else g.to(BF16).float()         # what the fp32-storage chain would see had g been rounded
        dc, dres = kk.bn_bwd(gin, None, xs[0], saved[0], True, dg, db, want_gout=True, ext=post.ext(0), dx_dtype=BF16)
        assert dres is gin
        out.append((g, post.parts, dc, dg, db))
    assert torch.equal(out[1][0], out[0][0].to(BF16))
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b), "the sums are those of the unrounded masked gradient"
    for a, b in zip(out[0][2:], out[1][2:]):
        assert torch.equal(a, b)
    # without the fused epilogue the gradient is not a block-output gradient: fp32 storage, whatever was asked
    plain = kk.conv2d_dgrad(dy, w, shape, None, 1, 0, 1, precision="bf16", out_dtype=BF16)
    assert plain.dtype == torch.float32


@pytest.mark.parametrize("shape", [(3, 20, 19, 128), (2, 28, 28, 256), (5, 9, 11, 512)])
def test_split_attention_passes_typed_storage(shape):
    """The four passes that read the radix convolution's raw output x0 (GAP, weighted sum, d(attention) + statistics, bn0
    backward) on a bf16-stored x0 == the same passes on the widened tensor; the weighted sum stored as bf16 == its RNE."""
    kk = K()
    B, H, W, C2 = shape
    Cp = C2 // 2
    rng = np.random.default_rng(sum(shape))
    x0 = _rand(rng, *shape).to(BF16)
    gamma, beta = _rand(rng, C2).abs() + 0.5, _rand(rng, C2)
    bn = kk.bn_stats(x0, gamma, beta, torch.zeros(C2, device="cuda"), torch.ones(C2, device="cuda"), True,
                     stats=_stats_of(x0.float()))
    a = torch.softmax(_rand(rng, B, 2, Cp), dim=1).reshape(B, C2).contiguous()
    dout, dgap = _rand(rng, B, H, W, Cp), _rand(rng, B, Cp)
    res = []
    for x in (x0.float(), x0):
        gap = kk.sa_gap(x, bn)
        out = kk.sa_apply_fwd(x, a, bn)
        da, sums = kk.sa_dattn(x, dout, bn, want_stats=True)
        da2 = kk.sa_dattn(x, dout, bn)
        dg, db = torch.zeros(C2, device="cuda"), torch.zeros(C2, device="cuda")
        dx = kk.sa_bn_bwd(dout, a, dgap, x, bn, True, dg, db, sums=sums)
        dg2, db2 = torch.zeros(C2, device="cuda"), torch.zeros(C2, device="cuda")
        dx2 = kk.sa_bn_bwd(dout, a, dgap, x, bn, True, dg2, db2)                     # own reduction pass
        dxp = kk.sa_bn_bwd(dout, a, dgap, x, bn, True, None, None, planes=1, keep_f32=False, sums=sums)
        res.append((gap, out, da, sums, da2, dx, dg, db, dx2, dg2, db2, dxp.planes))
    for i, (p, q) in enumerate(zip(*res)):
        assert torch.equal(p, q), i
    out16 = kk.sa_apply_fwd(x0, a, bn, out_dtype=BF16)
    assert out16.dtype == BF16 and torch.equal(out16, res[0][1].to(BF16))


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 128, 2), (3, 14, 14, 256, 512, 2), (2, 28, 28, 128, 256, 2)])
def test_plane_forward_stores_bf16(case):
    """One-plane plane convolution (the bf16 mode's 3x3 layers) writing its output as bf16 == RNE of the fp32-storage launch,
    same fused statistics, for every tile that supports it; the persistent tile 6 refuses."""
    B, H, W, Cin, Cout, groups = case
    kk = K()
    rng = np.random.default_rng(sum(case))
    xp = _rand(rng, B, H, W, Cin).to(BF16).unsqueeze(0).contiguous()
    w = _rand(rng, 3, 3, Cin // groups, Cout, scale=0.1)
    wf, _ = kk.planes_split_weight(w, groups, 1, fwd=True, dgrad=False)
    for tile in kk._plane_tiles(Cout // groups, 1, False) + (5,):
        if tile == 6:
            with pytest.raises(RuntimeError, match="tile 6"):
                kk._native.check(kk._native.lib().scouter_conv2d_fwd_planes_io(
                    xp.data_ptr(), wf.data_ptr(), None, None, xp.data_ptr(), None, B, H, W, Cin, Cout, 3, 3, 1, 1, groups, 0, 1,
                    6, kk.IO_Y_BF16, None), "conv2d_fwd_planes")
            continue
        if tile == 5 and not kk._halo_ok(3, 3, 1, 1, H, W, 1):
            continue
        y32, (p32, _) = kk.conv2d_fwd_planes(xp, wf, 3, 3, 1, 1, groups, bn_stats=True, tile=tile)
        y16, (p16, _) = kk.conv2d_fwd_planes(xp, wf, 3, 3, 1, 1, groups, bn_stats=True, tile=tile, out_dtype=BF16)
        assert y16.dtype == BF16 and torch.equal(y16, y32.to(BF16)) and torch.equal(p16, p32), "tile %d" % tile


def test_eval_forward_and_storage_switch_at_model_level(monkeypatch):
    """resnest26d in precision bf16, eval mode (running statistics: the BatchNorm passes read bf16-stored inputs without
    producer statistics): the forward with bf16 storage stays within bf16 noise of the fp32-storage forward, differs from it
    (proof that the typed kernels ran), and switching the storage back reproduces the fp32-storage bits."""
    import test_model_gpu as T
    monkeypatch.undo()
    m, P, images, labels, cfg = T._synthetic_model("resnest26d", 10, 1, 3, 8, 224, 2100)
    m.set_precision("bf16")
    m.eval()
    with torch.no_grad():
        o_bf = m(images.cuda(), labels.cuda())[0].clone()
        m.set_activation_storage("fp32")
        o_f32 = m(images.cuda(), labels.cuda())[0].clone()
        m.set_activation_storage("bf16")
        o_bf2 = m(images.cuda(), labels.cuda())[0].clone()
        m.set_activation_storage("fp32")
        o_f32b = m(images.cuda(), labels.cuda())[0].clone()
    assert torch.isfinite(o_bf).all()
    assert torch.equal(o_bf, o_bf2) and torch.equal(o_f32, o_f32b)
    d = float((o_bf - o_f32).abs().max())
    assert 0.0 < d < 0.5, d
    with pytest.raises(ValueError, match="precision 'bf16'"):
        m.set_precision("fp32")
        m.set_activation_storage("bf16")


def test_stem_pool_backward_stores_bf16():
    """fused BatchNorm + ReLU + max-pool backward writing dx as bf16 == RNE of the fp32-storage launch, same dgamma / dbeta"""
    kk = K()
    rng = np.random.default_rng(11)
    B, H, W, C = 3, 30, 34, 64
    x = _rand(rng, B, H, W, C)
    gamma, beta = _rand(rng, C).abs() + 0.5, _rand(rng, C)
    saved = kk.bn_stats(x, gamma, beta, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True)
    p, arg = kk.bn_maxpool_fwd(x, saved, 3, 2, 1, want_argmax=True)
    dy = _rand(rng, *p.shape)
    outs = []
    for dt in (torch.float32, BF16):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        outs.append((kk.bn_maxpool_bwd(dy, arg, x, saved, True, dg, db, 3, 2, 1, dx_dtype=dt), dg, db))
    assert outs[1][0].dtype == BF16 and torch.equal(outs[1][0], outs[0][0].to(BF16))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_residual_stream_gradient_switch(monkeypatch):
    """SCOUTER_BF16_GRAD_STREAM (resnest.GRAD_STREAM_BF16_DEFAULT -> ResNestBottleneck.grad_stream_bf16, set with the
    activation storage by SlotModel.set_activation_storage): with the masked block-output gradient kept in fp32 the step still
    runs on the typed kernels, the forward is the same bits, and the parameter gradients differ from the default by a
    bf16-rounding-sized amount only (one rounding per block of a tensor that fp32 arithmetic reads)."""
    import test_model_gpu as T
    from scouter_amd.timm.models import resnest as R
    monkeypatch.undo()
    res = []
    for flag in (True, False):
        monkeypatch.setattr(R, "GRAD_STREAM_BF16_DEFAULT", flag)
        m, P, images, labels, cfg = T._synthetic_model("resnest26d", 10, 1, 3, 8, 224, 2300)
        m.set_precision("bf16")
        out, (loss, nll, area) = m(images.cuda(), labels.cuda())
        loss.backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), m.grad_arena().flat.double().clone()))
    assert torch.equal(res[0][0], res[1][0])
    rel = float((res[0][1] - res[1][1]).norm() / res[1][1].norm())
    assert 0.0 < rel < 2e-2, rel
