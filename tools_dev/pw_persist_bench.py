"""The persistent pointwise kernel (tile 4, csrc/conv_pw_persist.h) against the implicit-GEMM tiles 0-3 on the short-K 1x1
layers of the benchmark step: forward (+ fused BatchNorm statistics) and plain input gradient.
usage: python tools_dev/pw_persist_bench.py [B]"""
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70


def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, H in [(64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (256, 512, 28), (256, 1024, 14)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    dy = torch.randn(B, H, H, cout, device='cuda')
    M = B * H * H
    fl = 2.0 * M * cin * cout
    byts = 4.0 * M * (cin + cout)
    res = {}
    for mode in ("fwd", "dgrad"):
        key = (mode, False, B, H, H, cin, cout, 1, 1, 1, 0, 1)
        for t in (0, 1, 2, 3, 4):
            legal = K._pw_persist_legal(M, cin if mode == "fwd" else cout, cout if mode == "fwd" else cin, 1, 1, 1, 0, 1, True) \
                if t == 4 else K._tile_legal(cout if mode == "fwd" else cin, t)
            if not legal:
                continue
            K._tile_cache[key] = t
            if mode == "fwd":
                res[(mode, t)] = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True))
            else:
                res[(mode, t)] = timeit(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 0, 1))
        K._tile_cache.pop(key, None)
    for mode in ("fwd", "dgrad"):
        old = min((v, t) for (m, t), v in res.items() if m == mode and t != 4)
        new = res.get((mode, 4))
        print("%-5s %4d->%4d @%2d: MFMA floor %5.1f us, HBM floor (6.3 TB/s) %5.1f us | best tile %d %6.1f us | persistent %s"
              % (mode, cin, cout, H, fl / 157.3e6, byts / 6.3e6, old[1], old[0],
                 "%6.1f us (%.2fx)" % (new, old[0] / new) if new else "n/a"))
