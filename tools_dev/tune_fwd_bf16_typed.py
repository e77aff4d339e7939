"""Block tiles of the 1x1 forward convolutions of BASELINE configs[4] under bf16 STORAGE (conv1: bf16 block input -> fp32
output; conv3: bf16 attention-weighted sum -> bf16 output; downsample: fp32 pooled input -> bf16 output), fused statistics."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
BF, F32 = torch.bfloat16, torch.float32
def timeit(fn, n=8):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot_t = tot_b = 0.0
for H, cin, cout, xdt, odt, n in [(56, 64, 64, F32, F32, 1), (56, 64, 256, BF, BF, 3), (56, 256, 64, BF, F32, 2), (56, 256, 128, BF, F32, 1),
                                  (28, 128, 512, BF, BF, 4), (28, 512, 128, BF, F32, 3), (28, 512, 256, BF, F32, 1),
                                  (14, 256, 1024, BF, BF, 6), (14, 1024, 256, BF, F32, 5), (14, 1024, 512, BF, F32, 1),
                                  (7, 512, 2048, BF, BF, 2), (7, 2048, 512, BF, F32, 2), (56, 64, 256, F32, BF, 1), (28, 256, 512, F32, BF, 1),
                                  (14, 512, 1024, F32, BF, 1), (7, 1024, 2048, F32, BF, 1)]:
    x = torch.randn(B, H, H, cin, device='cuda').to(xdt)
    w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    key = ("fwd", True, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    old = K._table_choice(key, lambda t, dry=False: True)
    res = {}
    for t in (0, 1, 2, 3):
        if not K._tile_legal(cout, t): continue
        K._tile_cache[key] = t
        res[t] = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, False, bn_stats=True, precision="bf16", out_dtype=odt))
    best = min(res, key=res.get)
    gb = (x.element_size() * x.numel() + (2 if odt == BF else 4) * B * H * H * cout) / 1e9
    print('fwd 1x1 %4d->%4d @%2d x%d (%s->%s): %s | table %s | best %d = %.0f GB/s' % (
        cin, cout, H, n, "bf16" if xdt == BF else "fp32", "bf16" if odt == BF else "fp32",
        "  ".join("%d: %.0f us" % kv for kv in sorted(res.items())), old, best, gb / res[best] * 1e6))
    tot_t += n * res.get(old, res[best]); tot_b += n * res[best]
print("per step: table %.2f ms -> best %.2f ms" % (tot_t / 1e3, tot_b / 1e3))
