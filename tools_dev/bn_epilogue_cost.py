"""Dev check: what the fused BatchNorm-statistics epilogue costs the forward convolution (same layer with / without).
usage: python tools_dev/bn_epilogue_cost.py"""
import sys, time, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = 70
layers = [  # cin, cout, k, groups, H, stride, pad
    (64, 64, 1, 1, 56, 1, 0), (64, 256, 1, 1, 56, 1, 0), (256, 64, 1, 1, 56, 1, 0), (256, 128, 1, 1, 56, 1, 0),
    (128, 512, 1, 1, 28, 1, 0), (512, 256, 1, 1, 28, 1, 0), (1024, 512, 1, 1, 14, 1, 0), (128, 256, 3, 2, 56, 1, 1),
]
for cin, cout, k, g, H, s, p in layers:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    res = []
    for stats in (False, True):
        for _ in range(3): K.conv2d_fwd(x, w, None, None, s, p, g, False, stats)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): K.conv2d_fwd(x, w, None, None, s, p, g, False, stats)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20)
    Ho = (H + 2 * p - k) // s + 1
    fl = 2.0 * B * Ho * Ho * cout * (cin // g) * k * k
    print("cin %4d cout %4d k%d g%d H%3d: plain %7.1f us (%5.1f TF/s)  +stats %7.1f us (%5.1f TF/s)  +%4.1f%%" % (
        cin, cout, k, g, H, res[0] * 1e6, fl / res[0] / 1e12, res[1] * 1e6, fl / res[1] / 1e12, 100 * (res[1] / res[0] - 1)))
