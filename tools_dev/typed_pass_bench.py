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
for B, H, C in ((256, 56, 256), (256, 28, 512), (256, 14, 1024)):
    shape = (B, H, H, C); n = B * H * H * C
    x = torch.randn(*shape, device='cuda'); dy = torch.randn(*shape, device='cuda'); res = torch.randn(*shape, device='cuda')
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    xd = x.double().reshape(-1, C)
    st = (torch.stack([xd.sum(0), (xd * xd).sum(0)], 1).unsqueeze(0).contiguous(), 1)
    _, saved, mask = K.bn_fwd(x, g, b, rm, rv, True, True, stats=st, want_mask=True)
    part = torch.zeros(1, C, 2, dtype=torch.float64, device='cuda')
    for name, (xx, dd, rr, od) in (("fp32", (x, dy, res, torch.float32)), ("bf16", (x.to(BF), dy.to(BF), res.to(BF), BF))):
        t_f = timeit(lambda: K.bn_fwd(xx, g, b, rm, rv, True, True, rr, stats=st, want_mask=True, out_dtype=od))
        t_b = timeit(lambda: K.bn_bwd(dd, None, xx, saved, True, None, None, ext=(part, 1), dx_dtype=od))
        es = 4 if name == "fp32" else 2
        print("%s %s: bn_fwd(+res) %.0f us = %.0f GB/s | bn_bwd(ext) %.0f us = %.0f GB/s" % (
            shape, name, t_f, 3 * es * n / t_f / 1e3, t_b, 3 * es * n / t_b / 1e3))
