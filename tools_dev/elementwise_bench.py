"""Dev timing of the HBM-bound helper kernels at the benchmark shapes (B = 70, resnest26d @224): microseconds and GB/s of
algorithmic traffic.   usage: python tools_dev/elementwise_bench.py"""
import sys, time, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = 70
def t_of(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def report(name, t, nbytes):
    print("%-44s %8.1f us  %7.0f GB/s" % (name, t * 1e6, nbytes / t / 1e9))
x = torch.randn(B, 112, 112, 64, device='cuda')
y, arg = K.maxpool_fwd(x); dy = torch.randn_like(y)
report("maxpool_fwd 112^2 x64", t_of(lambda: K.maxpool_fwd(x)), 4 * (x.numel() + y.numel()) + y.numel())
report("maxpool_bwd 112^2 x64", t_of(lambda: K.maxpool_bwd(dy, arg, tuple(x.shape))), 4 * (x.numel() + y.numel()) + y.numel())
for H, C in ((56, 128), (28, 256), (14, 512)):       # avd pools of layers 2-4 (k3 s2 p1) and the avg-down shortcuts (k2 s2 ceil)
    x = torch.randn(B, H, H, C, device='cuda')
    for (k, s, p, ceil, cip) in ((3, 2, 1, False, True), (2, 2, 0, True, False)):
        y = K.avgpool_fwd(x, k, s, p, ceil, cip); dy = torch.randn_like(y)
        report("avgpool_fwd k%d %d^2 x%d" % (k, H, C), t_of(lambda: K.avgpool_fwd(x, k, s, p, ceil, cip)), 4 * (x.numel() + y.numel()))
        report("avgpool_bwd k%d %d^2 x%d" % (k, H, C), t_of(lambda: K.avgpool_bwd(dy, tuple(x.shape), k, s, p, ceil, cip)), 4 * (x.numel() + y.numel()))
img = torch.randn(B, 3, 224, 224, device='cuda')
col = K.im2col_nchw(img, 3, 2, 1, 32)
report("im2col_nchw 224^2 k3 s2 -> [M][32]", t_of(lambda: K.im2col_nchw(img, 3, 2, 1, 32)), 4 * (img.numel() + col.numel()))
from scouter_amd.optim import FusedAdamW
ps = [torch.nn.Parameter(torch.randn(n, device='cuda')) for n in [2048 * 512 * 2] * 6 + [512 * 512 * 9] * 2 + [65536] * 20 + [2048] * 60 + [64] * 100]
for p_ in ps: p_.grad = torch.randn_like(p_)
opt = FusedAdamW(ps, lr=1e-4)
ntot = sum(p_.numel() for p_ in ps)
report("fused AdamW, %.1f M parameters in %d tensors" % (ntot / 1e6, len(ps)), t_of(lambda: opt.step()), 4 * 7 * ntot)
