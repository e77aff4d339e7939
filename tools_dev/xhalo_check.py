"""csrc/conv_xhalo.hip (tile 7 of the register-split entry points) against the exact-fp32 MFMA kernels and an fp64 reference
(torch conv on the CPU for small shapes, sampled on the GPU for the large ones): forward (+ fused BatchNorm statistics), input
gradient plain / with the fused BatchNorm-backward epilogue; timings of both.   python tools_dev/xhalo_check.py [quick]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scouter_amd import kernels as K

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ref_fwd64(x, w, groups):
    # x NHWC fp32 (cuda), w HWIO; returns NHWC fp64 (on the GPU: fp64 conv via unfold-free torch conv2d in double)
    xd = x.double().permute(0, 3, 1, 2)
    wd = w.double().permute(3, 2, 0, 1)
    return torch.nn.functional.conv2d(xd, wd, None, 1, 1, 1, groups).permute(0, 2, 3, 1).contiguous()


def ref_dgrad64(dy, w, groups):
    dyd = dy.double().permute(0, 3, 1, 2)
    wd = w.double().permute(3, 2, 0, 1)
    return torch.nn.functional.conv_transpose2d(dyd, wd, None, 1, 1, 0, groups).permute(0, 2, 3, 1).contiguous()


def check_fwd(B, H, W, Cin, Cout, groups, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=g) / np.sqrt(9 * Cin // groups)
    wf, _ = K.planes_split_weight(w, groups, 3, fwd=True, dgrad=False)
    y32, (p32, r32) = K.conv2d_fwd(x, w, None, None, 1, 1, groups, False, True)
    yx, (px, rx) = K.conv2d_fwd_x3(x, wf, None, False, True, tile=7, kh=3, pad=1, groups=groups)
    big = B * H * W * Cin > 6e7
    if big:
        sl = slice(0, 2)
        ref = ref_fwd64(x[sl], w, groups)
        a32, ax = y32[sl], yx[sl]
    else:
        ref = ref_fwd64(x, w, groups); a32, ax = y32, yx
    e32 = float((a32.double() - ref).abs().max()); ex = float((ax.double() - ref).abs().max())
    s32, sx = p32[:r32].sum(0), px[:rx].sum(0)
    ds = float((s32 - sx).abs().max() / s32.abs().max())
    yd = yx.double().view(-1, Cout)
    own = float((sx[:, 0] - yd.sum(0)).abs().max() / yd.sum(0).abs().max()), float((sx[:, 1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).abs().max())
    t32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 1, groups, False, True))
    tx = timeit(lambda: K.conv2d_fwd_x3(x, wf, None, False, True, tile=7, kh=3, pad=1, groups=groups))
    fl = 2.0 * B * H * W * Cout * (Cin // groups) * 9
    print("fwd  %-28s |err| fp32 %.2e halo %.2e | stats vs fp32 kernel %.1e, vs own output %.1e %.1e | %7.1f us (%5.1f TF) -> %7.1f us (%5.1f TF)"
          % (str((B, H, W, Cin, Cout, groups)), e32, ex, ds, own[0], own[1], t32, fl / t32 / 1e6, tx, fl / tx / 1e6), flush=True)
    assert ex <= max(e32, 1e-6) * 1.2 and ds < 1e-5 and max(own) < 1e-9, "FORWARD MISMATCH"


def check_dgrad(B, H, W, Cin, Cout, groups, seed, fused, with_add=False):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    dy = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    w = torch.randn(3, 3, Cin // groups, Cout, device="cuda", generator=g) / np.sqrt(9 * Cout // groups)
    add = torch.randn(B, H, W, Cin, device="cuda", generator=g) if with_add else None
    xs = (B, H, W, Cin)
    _, wd = K.planes_split_weight(w, groups, 3, fwd=False, dgrad=True)
    post32 = postx = None
    if fused:
        x1 = torch.randn(*xs, device="cuda", generator=g)
        g_, b_ = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        o = K.bn_fwd(x1, g_, b_, torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda"), True, True, want_mask=True)
        saved, mask = o[1], o[2]
        mk = lambda: K.BnBwdFuse(mask, [(x1, saved)])
        post32, postx = mk(), mk()
    else:
        mk = lambda: None
    d32 = K.conv2d_dgrad(dy, w, xs, add, 1, 1, groups, post=post32)
    dx = K.conv2d_dgrad_x3(dy, wd, xs, add, post=postx, tile=7, kh=3, pad=1, groups=groups)
    sl = slice(0, 2) if B * H * W * Cout > 6e7 else slice(0, B)
    ref = ref_dgrad64(dy[sl], w, groups)
    if with_add:
        ref = ref + add[sl].double()
    a32, ax = d32[sl].double(), dx[sl].double()
    if fused:
        assert post32.applied and postx.applied
        zero = (a32 == 0) | (ax == 0)
        nz = int(((a32 == 0) != (ax == 0)).sum())
        ref = torch.where(zero, torch.zeros_like(ref), ref); a32 = torch.where(zero, torch.zeros_like(a32), a32); ax = torch.where(zero, torch.zeros_like(ax), ax)
        s32 = [p[:post32.rows].sum(0) for p in post32.parts]; sx = [p[:postx.rows].sum(0) for p in postx.parts]
        dsum = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(s32, sx))
    else:
        nz, dsum = 0, 0.0
    e32 = float((a32 - ref).abs().max()); ex = float((ax - ref).abs().max())
    t32 = timeit(lambda: K.conv2d_dgrad(dy, w, xs, add, 1, 1, groups, post=mk()))
    tx = timeit(lambda: K.conv2d_dgrad_x3(dy, wd, xs, add, post=mk(), tile=7, kh=3, pad=1, groups=groups))
    fl = 2.0 * B * H * W * Cin * (Cout // groups) * 9
    print("dgrad%s %-28s |err| fp32 %.2e halo %.2e | mask mismatches %d, sums rel diff %.1e | %7.1f us (%5.1f TF) -> %7.1f us (%5.1f TF)"
          % ("+bn" if fused else "   ", str((B, H, W, Cin, Cout, groups)), e32, ex, nz, dsum, t32, fl / t32 / 1e6, tx, fl / tx / 1e6), flush=True)
    assert ex <= max(e32, 1e-6) * 1.2 and nz <= 2 and dsum < 1e-4, "DGRAD MISMATCH"


if __name__ == "__main__":
    import __graft_entry__ as G
    G.build()
    small = [(3, 9, 7, 32, 32, 1), (2, 20, 19, 64, 64, 2), (1, 5, 126, 32, 32, 1), (5, 33, 40, 96, 96, 3), (37, 14, 14, 32, 32, 1)]
    for i, c in enumerate(small):
        check_fwd(*c, seed=10 + i)
    smalld = [(3, 9, 7, 32, 32, 1), (2, 20, 19, 64, 128, 2), (1, 5, 126, 32, 64, 1), (5, 33, 40, 96, 192, 3), (37, 14, 14, 32, 32, 1)]
    for i, c in enumerate(smalld):
        check_dgrad(*c, seed=20 + i, fused=False, with_add=(i % 2 == 1))
        check_dgrad(*c, seed=30 + i, fused=True, with_add=(i % 2 == 0))
    for B in (6, 4):                  # the parity fixtures' batch: 294 tiles on 256 workgroups (two rounds), full comparison
        check_fwd(B, 112, 112, 32, 32, 1, 50 + B)
        check_dgrad(B, 112, 112, 32, 32, 1, 51 + B, fused=True)
        check_dgrad(B, 112, 112, 32, 64, 1, 52 + B, fused=True)
        check_dgrad(B, 56, 56, 64, 128, 2, 53 + B, fused=True)
    if not quick:
        B = 70
        check_fwd(B, 112, 112, 32, 32, 1, 40)
        check_dgrad(B, 112, 112, 32, 32, 1, 41, fused=True)
        check_dgrad(B, 112, 112, 32, 64, 1, 42, fused=True)
        check_dgrad(B, 56, 56, 64, 128, 2, 43, fused=True)
        check_dgrad(B, 56, 56, 64, 128, 2, 44, fused=False)
    print("XHALO_OK")
