"""1x1 weight gradients of resnest26d's deep layers: exact-fp32 MFMA kernel (the table's plan) vs the register-split bf16x3
kernel (csrc/conv_x3.hip xwgrad_kernel) per split-K plan.  usage: python tools_dev/xwgrad_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
shapes = [(128, 512, 28, 2), (256, 512, 28, 1), (512, 128, 28, 1), (512, 256, 28, 1), (256, 1024, 14, 2), (512, 1024, 14, 1),
          (1024, 256, 14, 1), (1024, 512, 14, 1), (512, 2048, 7, 2), (1024, 2048, 7, 1), (2048, 512, 7, 1)]
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t32 = tx = 0.0
for cin, cout, H, cnt in shapes:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    dw = torch.empty(1, 1, cin, cout, device='cuda')
    fl = 2.0 * B * H * H * cin * cout
    a = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 1, 0, 1))
    r = {p: timeit(lambda: K.conv2d_wgrad_x3(x, dy, dw, plan=p)) for p in K._X3_WGRAD_PLANS}
    b = min(r.values())
    print("%-18s %5.1f GF | fp32 %7.1f us (%5.1f TF) | x3 %s | best %.1f (%5.1f TF, %.2fx)" % (
        str((cin, cout, H)), fl / 1e9, a, fl / a / 1e6, " ".join("%d:%6.1f" % kv for kv in sorted(r.items())), b, fl / b / 1e6, a / b))
    t32 += a * cnt; tx += min(a, b) * cnt
print("per step: fp32 %.0f us -> best of both %.0f us" % (t32, tx))
