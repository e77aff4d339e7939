"""Plain pointwise input gradients of resnest50d under bf16 gradient storage (BASELINE configs[4]; dy bf16, dx fp32): the tile
kernels (igemm_bf16_kernel, table tile) vs the persistent typed kernel (tile 4, csrc/conv_pw_persist_bf16.h pwb_dgrad_kernel).
usage: python tools_dev/pwb_dgrad_bench.py [B]"""
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
# cin (dx), cout (dy), H, launches per step
for cin, cout, H, cnt in [(64, 256, 56, 4), (64, 64, 56, 1), (128, 512, 28, 4), (256, 512, 28, 1), (256, 1024, 14, 6), (512, 1024, 14, 1)]:
    dy = torch.randn(B, H, H, cout, device='cuda').to(BF); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    shape = (B, H, H, cin)
    t = []
    for on in (False, True):
        K.PWB_DGRAD = on
        t.append(timeit(lambda: K.conv2d_dgrad(dy, w, shape, None, 1, 0, 1, precision="bf16")))
    mb = B * H * H * (cout * 2 + cin * 4) / 1e6
    print("%-18s %6.0f MB | tile kernel %6.1f us (%.2f TB/s) | persistent %6.1f us (%.2f TB/s)" % (str((cin, cout, H)), mb, t[0], mb / t[0], t[1], mb / t[1]))
    tot[0] += cnt * t[0]; tot[1] += cnt * min(t)
print("per step: %.0f -> %.0f us" % tuple(tot))
