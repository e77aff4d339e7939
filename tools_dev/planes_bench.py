"""Plane convolution (bf16x3 / bf16) vs the fp32-MFMA kernel on the 3x3 layer shapes of resnest26d, batch 70."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
shapes = [(128, 256, 3, 2, 56), (256, 512, 3, 2, 28), (512, 1024, 3, 2, 14), (64, 128, 3, 2, 56), (128, 256, 3, 2, 28),
          (256, 512, 3, 2, 14), (512, 1024, 3, 2, 7), (32, 64, 3, 1, 112), (256, 1024, 1, 1, 14), (512, 2048, 1, 1, 7)]
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("%-28s %8s | %9s %9s %9s | %9s %9s" % ("cin,cout,k,g,H", "GFLOP", "fp32 us", "x3 us", "bf16 us", "x3 TF/s", "dgrad x3"))
for cin, cout, k, g, H in shapes:
    p = k // 2
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    fl = 2.0 * B * H * H * cout * (cin // g) * k * k
    t32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, p, g, bn_stats=True))
    xp3, xp1 = K.planes_split(x, 3), K.planes_split(x, 1)
    wf3, wd3 = K.planes_split_weight(w, g, 3); wf1, _ = K.planes_split_weight(w, g, 1)
    t3 = [timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, k, k, 1, p, g, bn_stats=False, tile=t)) for t in ((0, 1, 2, 3, 4) if (cout // g) % 128 == 0 else (1, 2, 3)) + ((5,) if k == 3 and H <= 63 else ()) + (6,)]; best3 = min(t3)
    best1 = min(timeit(lambda: K.conv2d_fwd_planes(xp1, wf1, k, k, 1, p, g, bn_stats=True, tile=t)) for t in (1,))
    td = float('nan')
    if (cin // g) % 64 == 0:
        y = K.conv2d_fwd(x, w, None, None, 1, p, g); dyp = K.planes_split(torch.randn_like(y), 3)
        tds = [timeit(lambda: K.conv2d_dgrad_planes(dyp, wd3, tuple(x.shape), k, k, 1, p, g, tile=t)) for t in ((0, 2, 4) if (cin // g) % 128 == 0 else (2, 3)) + ((5,) if k == 3 and H <= 63 else ()) + (6,)]
        td = min(tds); print('   dgrad x3 by tile', ['%.0f' % v for v in tds])
    tw32 = tw3 = float('nan')
    if (cin // g) % 64 == 0 and (cout // g) % 64 == 0:
        y = K.conv2d_fwd(x, w, None, None, 1, p, g); dyt = torch.randn_like(y); dwb = torch.empty_like(w)
        dyp3 = K.planes_split(dyt, 3)
        tw32 = timeit(lambda: K.conv2d_wgrad(x, dyt, dwb, 1, p, g)); tw3 = timeit(lambda: K.conv2d_wgrad_planes(xp3, dyp3, dwb, p, g))
    print("   wgrad fp32 %.1f us, bf16x3 %.1f us (%.1f TF/s)" % (tw32, tw3, fl / tw3 / 1e6))
    print("%-28s %8.2f | %9.1f %9.1f %9.1f | %9.1f %9.1f" % (str((cin, cout, k, g, H)), fl / 1e9, t32, best3, best1, fl / best3 / 1e6, fl / td / 1e6), ["%.0f" % v for v in t3])

