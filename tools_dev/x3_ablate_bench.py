"""Times the register-split GEMM (tile 0 = 256x128 and tile 2 = 128x64) on four 1x1 layers at batch 70; run once per ablated
library (tools_dev/x3_ablate.sh).  Prints us per launch."""
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
for cin, cout, H in [(1024, 512, 14), (512, 1024, 14), (256, 512, 28), (128, 512, 28), (1024, 2048, 7)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    wf, wd = K.planes_split_weight(w, 1, 3)
    out.append("%5.1f/%5.1f" % (timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=0)), timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=2))))
print("%-34s tile0/tile2 us: %s" % (os.environ.get("SCOUTER_HIP_LIB", "product"), "  ".join(out)))
