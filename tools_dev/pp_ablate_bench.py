"""Times plane tile 6 (persistent) and tile 4 on the three big grouped 3x3 layers and two deep 1x1 layers (batch 70);
run once per ablated library (tools_dev/pp_ablate.sh)."""
import os, sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = 70
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for cin, cout, k, g, H in [(128, 256, 3, 2, 56), (256, 512, 3, 2, 28), (512, 1024, 3, 2, 14), (1024, 2048, 1, 1, 7), (512, 1024, 1, 1, 14)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    xp3 = K.planes_split(x, 3); wf3, wd3 = K.planes_split_weight(w, g, 3)
    out.append("%.0f/%.0f" % (timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, k, k, 1, k // 2, g, tile=6)),
                              timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, k, k, 1, k // 2, g, tile=4))))
print(os.environ.get("SCOUTER_HIP_LIB", "product"), "tile6/tile4 us:", out)
