"""bf16-mode weight gradient of the 32-channel-group layers of resnest50d at batch 256 (BASELINE configs[4]): the fp32
tap-fused kernel they used before vs the bf16 kernel's ragged 64-wide tiles, every plan."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K

def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for H, cin, cout, g in [(112, 32, 32, 1), (112, 32, 64, 1), (56, 64, 128, 2)]:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    dw = torch.empty(3, 3, cin // g, cout, device='cuda')
    res = {}
    for prec in ("fp32", "bf16"):
        key = ("wgrad", prec == "bf16", B, H, H, cin, cout, 3, 3, 1, 1, g)
        for p in K._WGRAD_PLANS:
            K._tile_cache[key] = p
            try:
                res[(prec, p)] = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 1, 1, g, precision=prec))
            except RuntimeError as e:
                res[(prec, p)] = float('inf')
    bf = min((v, k[1]) for k, v in res.items() if k[0] == "bf16")
    fp = min((v, k[1]) for k, v in res.items() if k[0] == "fp32")
    print("wgrad 3x3 %d->%d g%d @%d B=%d: fp32 best %.1f us (plan %d, static %.1f) | bf16 best %.1f us (plan %d, static %.1f)"
          % (cin, cout, g, H, B, fp[0], fp[1], res[("fp32", -1)], bf[0], bf[1], res[("bf16", -1)]))
