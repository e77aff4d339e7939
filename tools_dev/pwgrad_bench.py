"""Plane weight gradient per (tile, split-K) plan on the 3x3 layer shapes of a backbone: the per-tap kernel (plans -1, 0..51)
vs the TAP-FUSED kernel (plans 64..67, csrc/conv_planes_wgrad_taps.h).  usage: python tools_dev/pwgrad_bench.py [B] [nplanes]"""
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shapes = [(128, 256, 2, 56), (128, 256, 2, 28), (256, 512, 2, 28), (256, 512, 2, 14), (512, 1024, 2, 14), (512, 1024, 2, 7)]


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, g, H in shapes:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    xp, dyp = K.planes_split(x, NP), K.planes_split(dy, NP)
    dw = torch.empty(3, 3, cin // g, cout, device='cuda')
    fl = 2.0 * B * H * H * cout * (cin // g) * 9
    key = ("pwgrad", NP, B, H, H, cin, cout, 3, 3, 1, g)
    res = {}
    for plan in K._PWGRAD_PLANS:
        K._tile_cache[key] = plan
        res[plan] = timeit(lambda: K.conv2d_wgrad_planes(xp, dyp, dw, 1, g))
    K._tile_cache.pop(key, None)
    old = min((v, k) for k, v in res.items() if k < 64)
    new = min((v, k) for k, v in res.items() if k >= 64)
    print("%-22s %6.2f GF | per-tap best plan %3d: %7.1f us %6.1f TF/s | tap-fused %s -> best %d: %7.1f us %6.1f TF/s"
          % ((cin, cout, g, H), fl / 1e9, old[1], old[0], fl / old[0] / 1e6,
             " ".join("%d:%.0f" % (k, v) for k, v in res.items() if k >= 64), new[1], new[0], fl / new[0] / 1e6))
