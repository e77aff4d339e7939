import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 70
for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
    x = torch.randn(B, H, H, C, device='cuda'); n = x.numel()
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    xd = x.double().reshape(-1, C)
    st = (torch.stack([xd.sum(0), (xd * xd).sum(0)], 1).unsqueeze(0).contiguous(), 1)
    t3 = timeit(lambda: K.bn_fwd(x, g, b, rm, rv, True, True, stats=st, want_mask=True, planes=3, keep_f32=False))
    t0 = timeit(lambda: K.bn_fwd(x, g, b, rm, rv, True, True, stats=st, want_mask=True))
    # sa_bn_bwd -> 3 planes
    C2 = 2 * C; x0 = torch.randn(B, H, H, C2, device='cuda')
    bn = K.bn_stats(x0, torch.ones(C2, device='cuda'), torch.zeros(C2, device='cuda'), torch.zeros(C2, device='cuda'), torch.ones(C2, device='cuda'), True)
    a = torch.softmax(torch.randn(B, 2, C, device='cuda'), 1).reshape(B, C2).contiguous()
    dout, dgap = torch.randn(B, H, H, C, device='cuda'), torch.randn(B, C, device='cuda')
    _, sums = K.sa_dattn(x0, dout, bn, want_stats=True)
    ts = timeit(lambda: K.sa_bn_bwd(dout, a, dgap, x0, bn, True, None, None, planes=3, keep_f32=False, sums=sums))
    print("@%d C=%d: bn1 apply -> 3 planes %.1f us (%.0f GB/s; + finalize launch) | fp32 out %.1f us (%.0f) | bn0 bwd -> 3 planes %.1f us (%.0f GB/s)" % (
        H, C, t3, 10 * n / t3 / 1e3, t0, 8 * n / t0 / 1e3, ts, (4 * x0.numel() + 4 * n + 6 * x0.numel()) / ts / 1e3))
