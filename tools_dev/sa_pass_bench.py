"""The four split-attention passes over the radix convolution's raw output x0 at BASELINE configs[4]'s sizes (batch 256),
fp32 vs bf16 storage of x0: time and algorithmic GB/s."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
BF = torch.bfloat16
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for H, C2 in ((56, 128), (28, 256), (14, 512), (7, 1024)):
    Cp = C2 // 2
    x32 = torch.randn(B, H, H, C2, device='cuda')
    g, b = torch.ones(C2, device='cuda'), torch.zeros(C2, device='cuda')
    bn = K.bn_stats(x32, g, b, torch.zeros(C2, device='cuda'), torch.ones(C2, device='cuda'), True)
    a = torch.softmax(torch.randn(B, 2, Cp, device='cuda'), 1).reshape(B, C2).contiguous()
    dout, dgap = torch.randn(B, H, H, Cp, device='cuda'), torch.randn(B, Cp, device='cuda')
    n2, n1 = x32.numel(), dout.numel()
    for name, x, od in (("fp32", x32, torch.float32), ("bf16", x32.to(BF), BF)):
        es = x.element_size()
        t_gap = timeit(lambda: K.sa_gap(x, bn))
        t_app = timeit(lambda: K.sa_apply_fwd(x, a, bn, out_dtype=od))
        t_dat = timeit(lambda: K.sa_dattn(x, dout, bn, want_stats=True))
        _, sums = K.sa_dattn(x, dout, bn, want_stats=True)
        t_bwd = timeit(lambda: K.sa_bn_bwd(dout, a, dgap, x, bn, True, None, None, planes=1, keep_f32=False, sums=sums))
        print("@%d C2=%d %s: gap %.0f us (%.0f GB/s) | apply %.0f us (%.0f) | dattn+stats %.0f us (%.0f) | bn0 bwd -> plane %.0f us (%.0f)" % (
            H, C2, name, t_gap, es * n2 / t_gap / 1e3, t_app, (es * n2 + es * n1) / t_app / 1e3,
            t_dat, (es * n2 + 4 * n1) / t_dat / 1e3, t_bwd, (es * n2 + 4 * n1 + 2 * n2) / t_bwd / 1e3))
