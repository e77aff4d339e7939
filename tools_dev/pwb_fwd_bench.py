"""Pointwise layers of resnest50d under bf16 storage (BASELINE configs[4]), forward with fused BatchNorm statistics: the tile
kernels (igemm_bf16_kernel, table tile) vs the persistent typed kernel (tile 4, csrc/conv_pw_persist_bf16.h pwb_fwd_kernel).
usage: python tools_dev/pwb_fwd_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF = torch.bfloat16
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0]
# cin, cout, H, output stored as bf16, launches per step
for cin, cout, H, ybf, cnt in [(64, 256, 56, True, 1), (256, 128, 56, False, 1), (256, 512, 28, True, 1), (512, 128, 28, False, 3), (512, 256, 28, False, 1),
                               (512, 1024, 14, True, 1), (256, 256, 56, False, 0), (128, 512, 28, True, 0), (256, 1024, 14, True, 0)]:
    x = torch.randn(B, H, H, cin, device='cuda').to(BF); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    od = BF if ybf else torch.float32
    t = []
    for on in (False, True):
        K.PWB_FWD = on
        t.append(timeit(lambda: K.conv2d_fwd(x, w, bn_stats=True, precision="bf16", out_dtype=od)))
    mb = B * H * H * (cin * 2 + cout * (2 if ybf else 4)) / 1e6
    print("%-22s %6.0f MB | tile kernel %6.1f us (%.2f TB/s) | persistent %6.1f us (%.2f TB/s)" % (str((cin, cout, H, 'y bf16' if ybf else 'y fp32')), mb, t[0], mb / t[0], t[1], mb / t[1]))
    tot[0] += cnt * t[0]; tot[1] += cnt * min(t)
print("listed launches per step: %.0f -> %.0f us" % tuple(tot))
