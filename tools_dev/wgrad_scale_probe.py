import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
def timeit(fn, n=30):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, H in [(512, 128, 28), (256, 1024, 14), (512, 2048, 7), (256, 64, 56)]:
    res = {}
    for B, plans in ((70, (49, 17, 33, 48, 50)), (210, (51, 50, 19, 35, 18, 34)), (140, (50, 51, 18, 34))):
        x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
        dw = torch.empty(1, 1, cin, cout, device='cuda')
        key = ("wgrad", False, B, H, H, cin, cout, 1, 1, 1, 0, 1)
        best = None
        for p in plans:
            K._tile_cache[key] = p
            t = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 1, 0, 1))
            best = t if best is None else min(best, t)
        res[B] = best
    print("wgrad %d->%d @%d: B=70 %.1f us | B=140 %.1f us (2 x B70 = %.1f) | B=210 %.1f us (3 x B70 = %.1f) -> fixed cost per launch ~ %.1f us"
          % (cin, cout, H, res[70], res[140], 2 * res[70], res[210], 3 * res[70], (3 * res[70] - res[210]) / 2))
