"""Short-K pointwise layers, forward with fused BatchNorm statistics: the fp32 kernels (table tile), the persistent bf16x3 kernel
(tile 5, csrc/conv_pw_persist_x3.h) and -- where kernels.x3_eligible -- the register-split GEMM's best tile.
usage: python tools_dev/xpw_fwd_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0]
for cin, cout, H, cnt in [(64, 64, 56, 1), (64, 256, 56, 3), (256, 64, 56, 1), (256, 128, 56, 1), (128, 512, 28, 2), (256, 512, 28, 1), (256, 1024, 14, 2),
                          (128, 256, 28, 0), (256, 256, 14, 0)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    t32 = timeit(lambda: K.conv2d_fwd(x, w, bn_stats=True))
    t5 = timeit(lambda: K.conv2d_fwd(x, w, bn_stats=True, tile=5))
    tx = float('nan')
    if K.x3_eligible(cin, cout, 1, 1, 1, 0, 1, False):
        wf, _ = K.planes_split_weight(w, 1, 3)
        tx = min(timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t)) for t in range(5) if K._x3_tile_ok(t, cout))
    y5 = K.conv2d_fwd(x, w, tile=5); y32 = K.conv2d_fwd(x, w)
    ref = (x.double().reshape(-1, cin) @ w.double().reshape(cin, cout))
    e5, e32 = float((y5.reshape(-1, cout).double() - ref).abs().max()), float((y32.reshape(-1, cout).double() - ref).abs().max())
    mb = B * H * H * (cin + cout) * 4 / 1e6
    print("%-16s %6.0f MB | fp32 %6.1f us | persistent bf16x3 %6.1f us (%.2f TB/s) | register-split GEMM %6.1f us | max err vs fp64: %.2e (fp32 kernel %.2e)"
          % (str((cin, cout, H)), mb, t32, t5, mb / t5, tx, e5, e32))
    cur = min(t32, tx) if tx == tx else t32
    tot[0] += cnt * cur; tot[1] += cnt * min(cur, t5)
print("per step: now %.0f us -> with the persistent kernel where it wins %.0f us" % tuple(tot))
