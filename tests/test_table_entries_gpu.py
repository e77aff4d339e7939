"""Parity at the kernel instances BASELINE configs[3] / configs[4] actually launch.

scouter_amd/tuning/gfx950.json names, per layer shape, the block tile / weight-gradient plan every process uses.  Its
B = 128 (CUB200 resnest26d) and B = 256 (ImageNet-100 resnest50d, fp32 and bf16) entries select large-M instances --
other split-K plans, plane tiles 5 / 6, other XCD remaps -- that the whole-model fixtures (B <= 70) never launch.  Here
EVERY such entry is launched at its real shape through the same wrappers the model uses (so the table lookup itself is
exercised: the test asserts the cached choice is the table's) and checked against exact fp64 arithmetic on a random
sample of output elements (a full fp64 convolution at 256 x 56 x 56 x 256 would take minutes per entry; a wrong tile
index, a dropped K-tile / tap / split-K slab or a mis-addressed remap corrupts whole tiles, which a 2048-element sample
cannot miss).  The fp64 sample is computed with torch indexing + einsum on the device: test infrastructure, independent
of the kernels under test.  bf16 entries are compared with the exact convolution of the RNE-rounded operands."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = json.load(open(os.path.join(ROOT, "scouter_amd", "tuning", "gfx950.json")))["choices"]
BATCHES = ("128", "256")
NS = 2048


def _entries(mode):
    out = []
    for ks, v in sorted(TABLE.items()):
        p = ks.split("|")
        if p[0] != mode:
            continue
        bp = 4 if mode in ("dgrad+bn", "pdgrad+bn") else 2
        if p[bp] in BATCHES:
            out.append(ks)
    return out


def K():
    from scouter_amd import kernels
    return kernels


def _rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, device="cuda") * scale).contiguous()


def _opnd(t, bf16):
    """what the kernel multiplies: the fp32 value, or its RNE bf16 rounding"""
    return (t.bfloat16() if bf16 else t).double()


def _sample_fwd(x, w, stride, pad, groups, gen, bf16, n=NS):
    """-> (index tuple into y [B,Ho,Wo,Cout], exact fp64 values, magnitude V = sqrt(sum of squared terms))"""
    B, H, W, Cin = x.shape
    kh, kw, cg, Cout = w.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    ri = lambda hi: torch.randint(0, hi, (n,), generator=gen, device="cuda")
    b, yo, xo, co = ri(B), ri(Ho), ri(Wo), ri(Cout)
    ng = Cout // groups
    c0 = (co // ng) * cg
    acc = torch.zeros(n, dtype=torch.float64, device="cuda")
    sq = torch.zeros_like(acc)
    ar = torch.arange(cg, device="cuda")
    for r in range(kh):
        for s in range(kw):
            yy, xx = yo * stride - pad + r, xo * stride - pad + s
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            xv = _opnd(x[b, yy.clamp(0, H - 1), xx.clamp(0, W - 1)], bf16)               # [n, Cin]
            xv = torch.gather(xv, 1, c0[:, None] + ar[None, :]) * ok[:, None]
            wv = _opnd(w[r, s][:, co].t(), bf16)                                          # [n, cg]
            t = xv * wv
            acc += t.sum(1)
            sq += (t * t).sum(1)
    return (b, yo, xo, co), acc, sq.sqrt()


def _sample_dgrad(dy, w, xshape, stride, pad, groups, gen, bf16, n=NS):
    B, H, W, Cin = xshape
    kh, kw, cg, Cout = w.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    ri = lambda hi: torch.randint(0, hi, (n,), generator=gen, device="cuda")
    b, y, x_, ci = ri(B), ri(H), ri(W), ri(Cin)
    ng = Cout // groups
    o0 = (ci // cg) * ng
    cil = ci % cg
    ar = torch.arange(ng, device="cuda")
    acc = torch.zeros(n, dtype=torch.float64, device="cuda")
    sq = torch.zeros_like(acc)
    for r in range(kh):
        for s in range(kw):
            ty, tx = y + pad - r, x_ + pad - s
            ok = (ty % stride == 0) & (tx % stride == 0)
            yo, xo = ty // stride, tx // stride
            ok &= (yo >= 0) & (yo < Ho) & (xo >= 0) & (xo < Wo)
            dv = _opnd(dy[b, yo.clamp(0, Ho - 1), xo.clamp(0, Wo - 1)], bf16)              # [n, Cout]
            dv = torch.gather(dv, 1, o0[:, None] + ar[None, :]) * ok[:, None]
            wv = torch.gather(_opnd(w[r, s][cil], bf16), 1, o0[:, None] + ar[None, :])     # [n, ng]
            t = dv * wv
            acc += t.sum(1)
            sq += (t * t).sum(1)
    return (b, y, x_, ci), acc, sq.sqrt()


def _sample_wgrad(x, dy, kh, kw, stride, pad, groups, gen, bf16, n=48):
    B, H, W, Cin = x.shape
    Ho, Wo, Cout = dy.shape[1], dy.shape[2], dy.shape[3]
    cg, ng = Cin // groups, Cout // groups
    idx, vals, mags = [], [], []
    for _ in range(n):
        r, s = int(torch.randint(0, kh, (1,), generator=gen, device="cuda")), int(torch.randint(0, kw, (1,), generator=gen, device="cuda"))
        cil = int(torch.randint(0, cg, (1,), generator=gen, device="cuda"))
        co = int(torch.randint(0, Cout, (1,), generator=gen, device="cuda"))
        ci = (co // ng) * cg + cil
        xs = torch.zeros((B, Ho, Wo), dtype=torch.float64, device="cuda")
        yo = torch.arange(Ho, device="cuda") * stride - pad + r
        xo = torch.arange(Wo, device="cuda") * stride - pad + s
        vy, vx = (yo >= 0) & (yo < H), (xo >= 0) & (xo < W)
        sub = _opnd(x[:, :, :, ci], bf16)[:, yo[vy]][:, :, xo[vx]]
        xs[:, vy.nonzero()[:, 0][:, None], vx.nonzero()[:, 0][None, :]] = sub
        t = xs * _opnd(dy[:, :, :, co], bf16)
        idx.append((r, s, cil, co)); vals.append(float(t.sum())); mags.append(float((t * t).sum().sqrt()))
    return idx, np.array(vals), np.array(mags)


def _check(name, got, ref, mag, k=2e-5):
    err = (got.double() - ref).abs()
    tol = k * mag + 1e-12
    bad = int((err > tol).sum())
    assert bad == 0, "%s: %d of %d sampled elements off, worst %.3g x its tolerance (|err| %.3g)" % (
        name, bad, err.numel(), float((err / tol).max()), float(err.max()))


def _chosen(kk, key, ks):
    """the wrapper cached the TABLE's choice for this key (when it is a legal candidate: else the library heuristic)"""
    assert key in kk._tile_cache, "the wrapper did not look up %s" % ks
    assert kk._tile_cache[key] in (TABLE[ks], -1), (ks, kk._tile_cache[key], TABLE[ks])
    return kk._tile_cache[key]


@pytest.fixture(autouse=True)
def _table_mode(monkeypatch):
    kk = K()
    monkeypatch.setattr(kk, "AUTOTUNE", "table")
    kk._tile_cache.clear()
    yield
    kk._tile_cache.clear()
    torch.cuda.empty_cache()


def _parse(ks, skip):
    p = ks.split("|")
    return [int(v) for v in p[skip:]]


@pytest.mark.parametrize("ks", _entries("fwd"))
def test_fp32_and_bf16_forward_entries(ks):
    bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    x = _rnd(gen, B, H, W, Cin)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cin // groups * kh * kw))
    y, (part, rows) = kk.conv2d_fwd(x, w, None, None, stride, pad, groups, False, True, precision="bf16" if bf16 else "fp32")
    key = ("fwd", bool(bf16), B, H, W, Cin, Cout, kh, kw, stride, pad, groups)
    _chosen(kk, key, ks)
    idx, ref, mag = _sample_fwd(x, w, stride, pad, groups, gen, bool(bf16))
    _check(ks, y[idx], ref, mag)
    # fused BatchNorm statistics of exactly this instance: fp64 sums of the kernel's own output
    yd = y.double().view(-1, Cout)
    s = part[:rows].sum(0)
    torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-9, atol=1e-7)
    torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("ks", _entries("dgrad"))
def test_fp32_and_bf16_input_gradient_entries(ks):
    bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    dy = _rnd(gen, B, Ho, Wo, Cout)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cout // groups * kh * kw))
    dx = kk.conv2d_dgrad(dy, w, (B, H, W, Cin), None, stride, pad, groups, precision="bf16" if bf16 else "fp32")
    _chosen(kk, ("dgrad", bool(bf16), B, H, W, Cin, Cout, kh, kw, stride, pad, groups), ks)
    idx, ref, mag = _sample_dgrad(dy, w, (B, H, W, Cin), stride, pad, groups, gen, bool(bf16))
    _check(ks, dx[idx], ref, mag)


@pytest.mark.parametrize("ks", _entries("wgrad"))
def test_fp32_and_bf16_weight_gradient_entries(ks):
    bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    x = _rnd(gen, B, H, W, Cin)
    dy = _rnd(gen, B, Ho, Wo, Cout)
    dw = torch.full((kh, kw, Cin // groups, Cout), float("nan"), device="cuda")
    kk.conv2d_wgrad(x, dy, dw, stride, pad, groups, precision="bf16" if bf16 else "fp32")
    # (the wrapper decides itself whether the bf16 kernel applies to this shape: the key carries its decision)
    keys = [k for k in kk._tile_cache if k[0] == "wgrad"]
    assert len(keys) == 1 and kk._key_str(keys[0]) == ks, (keys, ks)
    _chosen(kk, keys[0], ks)
    assert bool(torch.isfinite(dw).all())
    idx, ref, mag = _sample_wgrad(x, dy, kh, kw, stride, pad, groups, gen, bool(bf16))
    got = np.array([float(dw[i]) for i in idx])
    err = np.abs(got - ref)
    assert (err <= 2e-5 * mag + 1e-12).all(), (ks, float((err / (2e-5 * mag + 1e-12)).max()))
    dw2 = torch.empty_like(dw)
    kk.conv2d_wgrad(x, dy, dw2, stride, pad, groups, precision="bf16" if bf16 else "fp32")
    assert torch.equal(dw, dw2), "split-K plan %d is not deterministic" % TABLE[ks]


def _planes(kk, t, nplanes):
    return kk.planes_split(t, nplanes)


@pytest.mark.parametrize("ks", _entries("pfwd"))
def test_plane_forward_entries(ks):
    nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    x = _rnd(gen, B, H, W, Cin)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cin // groups * kh * kw))
    wf, _ = kk.planes_split_weight(w, groups, nplanes, fwd=True, dgrad=False)
    xp = _planes(kk, x, nplanes)
    key = ("pfwd", nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups)
    idx, ref, mag = _sample_fwd(x, w, stride, pad, groups, gen, nplanes == 1)

    def check(y, part, rows):
        _check(ks, y[idx], ref, mag)
        yd = y.double().view(-1, Cout)
        s = part[:rows].sum(0)
        torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-9, atol=1e-7)
        torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)
    y, (part, rows) = kk.conv2d_fwd_planes(xp, wf, kh, kw, stride, pad, groups, bn_stats=True)
    check(y, part, rows)
    if kk._halo_ok(kh, kw, stride, pad, H, W, 1) and H * W >= kk.HALO_FWD_MIN_PIXELS:
        # the FORWARD of same-size 3x3 layers on maps of at least 14 x 14 pixels runs the resident-rows tile by a STATIC rule
        # (kernels.HALO_TILE bit 0, default since round 6): no table lookup, the same bits for every batch / process / rank.
        # The table's entry is what the layer runs with SCOUTER_HALO=2 -- that instance is checked here as well.
        assert key not in kk._tile_cache, "the static forward rule must not consult the table (%s)" % ks
        y, (part, rows) = kk.conv2d_fwd_planes(xp, wf, kh, kw, stride, pad, groups, bn_stats=True, tile=TABLE[ks])
        check(y, part, rows)
    else:
        _chosen(kk, key, ks)


@pytest.mark.parametrize("ks", _entries("pdgrad"))
def test_plane_input_gradient_entries(ks):
    nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    dy = _rnd(gen, B, H, W, Cout)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cout // groups * kh * kw))
    _, wd = kk.planes_split_weight(w, groups, nplanes, fwd=False, dgrad=True)
    dx = kk.conv2d_dgrad_planes(_planes(kk, dy, nplanes), wd, (B, H, W, Cin), kh, kw, stride, pad, groups)
    _chosen(kk, ("pdgrad", nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), ks)
    idx, ref, mag = _sample_dgrad(dy, w, (B, H, W, Cin), stride, pad, groups, gen, nplanes == 1)
    _check(ks, dx[idx], ref, mag)


@pytest.mark.parametrize("ks", _entries("pwgrad"))
def test_plane_weight_gradient_entries(ks):
    nplanes, B, H, W, Cin, Cout, kh, kw, pad, groups = _parse(ks, 1)
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    x = _rnd(gen, B, H, W, Cin)
    dy = _rnd(gen, B, H, W, Cout)
    dw = torch.full((kh, kw, Cin // groups, Cout), float("nan"), device="cuda")
    xp, dyp = _planes(kk, x, nplanes), _planes(kk, dy, nplanes)
    kk.conv2d_wgrad_planes(xp, dyp, dw, pad, groups)
    _chosen(kk, ("pwgrad", nplanes, B, H, W, Cin, Cout, kh, kw, pad, groups), ks)
    assert bool(torch.isfinite(dw).all())
    idx, ref, mag = _sample_wgrad(x, dy, kh, kw, 1, pad, groups, gen, nplanes == 1)
    got = np.array([float(dw[i]) for i in idx])
    err = np.abs(got - ref)
    assert (err <= 2e-5 * mag + 1e-12).all(), (ks, float((err / (2e-5 * mag + 1e-12)).max()))
    dw2 = torch.empty_like(dw)
    kk.conv2d_wgrad_planes(xp, dyp, dw2, pad, groups)
    assert torch.equal(dw, dw2)


def _bn_setup(kk, gen, B, H, W, C, n, with_add):
    """the BatchNorm(s) whose output gradient the input-gradient kernel produces: forward first (mask, saved blocks)"""
    bns = []
    for i in range(n):
        x = _rnd(gen, B, H, W, C) * 1.7 + 0.4
        gamma, beta = torch.rand(C, generator=gen, device="cuda") + 0.5, _rnd(gen, C)
        bns.append((x, gamma, beta, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")))
    x1, g1, b1, rm1, rv1 = bns[0]
    if n == 2:
        x2, g2, b2, rm2, rv2 = bns[1]
        y2, saved2 = kk.bn_fwd(x2, g2, b2, rm2, rv2, True, False)
        _, saved1, mask = kk.bn_fwd(x1, g1, b1, rm1, rv1, True, True, residual=y2, want_mask=True)
        return mask, [(x1, saved1), (x2, saved2)]
    res = _rnd(gen, B, H, W, C)
    _, saved1, mask = kk.bn_fwd(x1, g1, b1, rm1, rv1, True, True, residual=res, want_mask=True)
    return mask, [(x1, saved1)]


def _check_fused(kk, ks, gf, plain, post, Cin):
    """fused launch == plain input gradient (sampled vs fp64 by the entries above) with the ReLU mask applied -- bit for
    bit when both ran the same tile -- and its fp64 partial sums == sums of that masked gradient"""
    C = Cin
    # (the mask is decoded by the BatchNorm backward itself rather than by re-stating its bit layout here)
    dgs = [torch.zeros(C, device="cuda") for _ in range(2)]
    x1, saved1 = post.entries[0]
    _, gout = kk.bn_bwd(plain, None, x1, saved1, True, dgs[0], dgs[1], True, mask=post.mask)
    err = float((gf - gout).abs().max())
    sc = float(gout.abs().max())
    assert err <= 4e-6 * sc, (ks, err, sc)                    # (other tile than the plain launch: K order may differ)
    gd = gf.double().view(-1, C)
    for i, (x, saved) in enumerate(post.entries):
        part, rows = post.ext(i)
        sums = part[:rows].sum(0)
        xhat = ((x.view(-1, C) - saved[0]) * saved[1]).double()
        torch.testing.assert_close(sums[:, 0], gd.sum(0), rtol=1e-9, atol=1e-7)
        torch.testing.assert_close(sums[:, 1], (gd * xhat).sum(0), rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("ks", _entries("dgrad+bn"))
def test_fused_input_gradient_entries(ks, monkeypatch):
    nent, with_add, bf16, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    dy = _rnd(gen, B, Ho, Wo, Cout)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cout // groups * kh * kw))
    addend = _rnd(gen, B, H, W, Cin) if with_add else None
    mask, entries = _bn_setup(kk, gen, B, H, W, Cin, nent, with_add)
    prec = "bf16" if bf16 else "fp32"
    post = kk.BnBwdFuse(mask, entries)
    gf = kk.conv2d_dgrad(dy, w, (B, H, W, Cin), addend, stride, pad, groups, precision=prec, post=post)
    assert post.applied
    _chosen(kk, ("dgrad+bn", nent, bool(with_add), bool(bf16), B, H, W, Cin, Cout, kh, kw, stride, pad, groups), ks)
    plain = kk.conv2d_dgrad(dy, w, (B, H, W, Cin), addend, stride, pad, groups, precision=prec)
    _check_fused(kk, ks, gf, plain, post, Cin)


@pytest.mark.parametrize("ks", _entries("pdgrad+bn"))
def test_fused_plane_input_gradient_entries(ks, monkeypatch):
    nent, with_add, nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups = _parse(ks, 1)
    kk = K()
    monkeypatch.setattr(kk, "BN_BWD_FUSE", 15)
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    dy = _rnd(gen, B, H, W, Cout)
    w = _rnd(gen, kh, kw, Cin // groups, Cout, scale=1.0 / np.sqrt(Cout // groups * kh * kw))
    addend = _rnd(gen, B, H, W, Cin) if with_add else None
    mask, entries = _bn_setup(kk, gen, B, H, W, Cin, nent, with_add)
    _, wd = kk.planes_split_weight(w, groups, nplanes, fwd=False, dgrad=True)
    dyp = _planes(kk, dy, nplanes)
    post = kk.BnBwdFuse(mask, entries)
    gf = kk.conv2d_dgrad_planes(dyp, wd, (B, H, W, Cin), kh, kw, stride, pad, groups, addend, post=post)
    if not post.applied:
        pytest.skip("this shape does not fuse in %d-plane mode (kernels.conv2d_dgrad_planes)" % nplanes)
    _chosen(kk, ("pdgrad+bn", nent, bool(with_add), nplanes, B, H, W, Cin, Cout, kh, kw, stride, pad, groups), ks)
    plain = kk.conv2d_dgrad_planes(dyp, wd, (B, H, W, Cin), kh, kw, stride, pad, groups, addend)
    _check_fused(kk, ks, gf, plain, post, Cin)
    idx, ref, mag = _sample_dgrad(dy, w, (B, H, W, Cin), stride, pad, groups, gen, nplanes == 1)
    if addend is not None:
        ref = ref + addend[idx].double()
    _check(ks + " (plain)", plain[idx], ref, mag + (addend[idx].abs().double() if addend is not None else 0))


# ---- the register-split bf16x3 kernels (csrc/conv_x3.hip, round 5): their table keys carry (nplanes = 3, B, H, W, Cin, Cout
# [, k, groups]); fp32 operands in, fp32-grade products -- same sampled fp64 check, fp32 tolerance
def _x_entries(mode):
    out = []
    for ks in sorted(TABLE):
        p = ks.split("|")
        if p[0] == mode and p[4 if mode == "xdgrad+bn" else 2] in BATCHES:
            out.append(ks)
    return out


@pytest.mark.parametrize("ks", _x_entries("xfwd"))
def test_register_split_forward_entries(ks):
    v = _parse(ks, 2)
    B, H, W, Cin, Cout = v[:5]
    kh, groups = (v[5], v[6]) if len(v) > 5 else (1, 1)
    pad = kh // 2
    kk = K()
    if not (kk.x3_eligible(Cin, Cout, kh, kh, 1, pad, groups, False) or kk.x3_conv_eligible(Cin, Cout, kh, kh, 1, pad, groups, False)):
        pytest.skip("shape no longer routed to the register-split kernel (kept in the table by an earlier tuning pass)")
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    x = _rnd(gen, B, H, W, Cin)
    w = _rnd(gen, kh, kh, Cin // groups, Cout, scale=1.0 / np.sqrt(Cin // groups * kh * kh))
    wf, _ = kk.planes_split_weight(w, groups, 3, fwd=True, dgrad=False)
    y, (part, rows) = kk.conv2d_fwd_x3(x, wf, bn_stats=True, kh=kh, pad=pad, groups=groups)
    key = ("xfwd", 3, B, H, W, Cin, Cout) if kh == 1 and groups == 1 else ("xfwd", 3, B, H, W, Cin, Cout, kh, groups)
    _chosen(kk, key, ks)
    idx, ref, mag = _sample_fwd(x, w, 1, pad, groups, gen, False)
    _check(ks, y[idx], ref, mag)
    yd = y.double().view(-1, Cout)
    s = part[:rows].sum(0)
    torch.testing.assert_close(s[:, 0], yd.sum(0), rtol=1e-9, atol=1e-7)
    torch.testing.assert_close(s[:, 1], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("ks", _x_entries("xdgrad"))
def test_register_split_input_gradient_entries(ks):
    B, H, W, Cin, Cout = _parse(ks, 2)[:5]
    kk = K()
    gen = torch.Generator(device="cuda"); gen.manual_seed(zlib.crc32(ks.encode()))
    dy = _rnd(gen, B, H, W, Cout)
    w = _rnd(gen, 1, 1, Cin, Cout, scale=1.0 / np.sqrt(Cout))
    add = _rnd(gen, B, H, W, Cin)
    _, wd = kk.planes_split_weight(w, 1, 3, fwd=False, dgrad=True)
    dx = kk.conv2d_dgrad_x3(dy, wd, (B, H, W, Cin), addend=add)
    _chosen(kk, ("xdgrad", 3, B, H, W, Cin, Cout), ks)
    idx, ref, mag = _sample_dgrad(dy, w, (B, H, W, Cin), 1, 0, 1, gen, False)
    _check(ks, dx[idx] - add[idx], ref, mag + add[idx].abs().double())
