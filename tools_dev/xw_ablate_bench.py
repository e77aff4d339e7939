"""Times the register-split weight gradient on four 1x1 layers at batch 70 (plan -1); run once per ablated library
(XW=1 bash tools_dev/x3_ablate.sh ...).  Prints us per launch (kernel + slab sum)."""
import os, sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = 70
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for cin, cout, H in [(512, 1024, 14), (256, 512, 28), (1024, 2048, 7), (512, 128, 28)]:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    dw = torch.empty(1, 1, cin, cout, device='cuda')
    out.append("%5.1f" % timeit(lambda: K.conv2d_wgrad_x3(x, dy, dw, plan=-1)))
print("%-34s us: %s" % (os.environ.get("SCOUTER_HIP_LIB", "product"), "  ".join(out)))
