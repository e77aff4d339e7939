"""Grouped 3x3 layers of resnest26d (cg >= 64): plane kernels (pre-split activation planes, every candidate tile) vs the
register-split kernel on the fp32 activation (csrc/conv_x3.hip, CONV = true) -- forward with fused statistics, plain input
gradient.  usage: python tools_dev/x3_conv_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
shapes = [(128, 256, 56), (128, 256, 28), (256, 512, 28), (256, 512, 14), (512, 1024, 14), (512, 1024, 7)]
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fm = lambda d: ' '.join('%d:%6.1f' % kv for kv in sorted(d.items()))
tp = tx = 0.0
for cin, cout, H in shapes:
    g, k, p = 2, 3, 1
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    fl = 2.0 * B * H * H * cout * (cin // g) * k * k
    xp = K.planes_split(x, 3); wf, wd = K.planes_split_weight(w, g, 3)
    ptiles = K._plane_tiles(cout // g, 3, False) + ((5,) if H <= 63 else ())
    fp = {t: timeit(lambda: K.conv2d_fwd_planes(xp, wf, k, k, 1, p, g, bn_stats=True, tile=t)) for t in ptiles}
    fx = {t: timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t, kh=3, pad=1, groups=g)) for t in range(5) if K._x3_tile_ok(t, cout // g)}
    dy = torch.randn(B, H, H, cout, device='cuda'); dyp = K.planes_split(dy, 3); xs = tuple(x.shape)
    dtiles = K._plane_tiles(cin // g, 3, H <= 63)
    dp = {t: timeit(lambda: K.conv2d_dgrad_planes(dyp, wd, xs, k, k, 1, p, g, tile=t)) for t in dtiles}
    dx = {t: timeit(lambda: K.conv2d_dgrad_x3(dy, wd, xs, tile=t, kh=3, pad=1, groups=g)) for t in range(5) if K._x3_tile_ok(t, cin // g)}
    a, b, c, d = min(fp.values()), min(fx.values()), min(dp.values()), min(dx.values())
    print("%-16s %5.1f GF | fwd planes %s | x3 %s | %.1f vs %.1f (%.0f / %.0f TF)" % (str((cin, cout, H)), fl / 1e9, fm(fp), fm(fx), a, b, fl / a / 1e6, fl / b / 1e6))
    print("%-16s          | dgrad planes %s | x3 %s | %.1f vs %.1f (%.0f / %.0f TF)" % ("", fm(dp), fm(dx), c, d, fl / c / 1e6, fl / d / 1e6))
    tp += a + c; tx += b + d
print("sum: planes %.0f us, x3 %.0f us" % (tp, tx))
