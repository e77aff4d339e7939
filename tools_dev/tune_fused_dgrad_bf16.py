"""Block tiles of the fused 1x1 input gradient (conv1 + the previous block's BatchNorm-backward sums) of BASELINE configs[4]
under bf16 STORAGE (dy, dx, addend, BatchNorm inputs all bf16): times tiles 0-3 and the persistent kernel (4) per shape,
prints the table lines."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
BF = torch.bfloat16

def timeit(fn, n=8):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot_old = tot_new = 0.0
# (H, Cin = block width 4g, Cout = g of the consuming conv1, BatchNorms finished, launches per step)
for H, cin, cout, nbn, n in [(56, 256, 64, 2, 1), (56, 256, 64, 1, 1), (56, 256, 128, 1, 1), (28, 512, 128, 2, 1), (28, 512, 128, 1, 2),
                             (28, 512, 256, 1, 1), (14, 1024, 256, 2, 1), (14, 1024, 256, 1, 4), (14, 1024, 512, 1, 1),
                             (7, 2048, 512, 2, 1), (7, 2048, 512, 1, 1)]:
    shape = (B, H, H, cin)
    dy = torch.randn(B, H, H, cout, device='cuda').to(BF)
    w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    add = torch.randn(*shape, device='cuda').to(BF)
    xs = [torch.randn(*shape, device='cuda').to(BF) for _ in range(nbn)]
    g, b = torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda')
    saved = [K.bn_stats(x.float(), g, b, torch.zeros(cin, device='cuda'), torch.ones(cin, device='cuda'), True) for x in xs]
    mask = torch.zeros(K._native.lib().scouter_relu_mask_words(xs[0].numel()), dtype=torch.int64, device='cuda')
    mask.random_()
    key = ("dgrad+bn", nbn, True, True, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    old = K._table_choice(key, lambda t, dry=False: True)
    res = {}
    for t in (0, 1, 2, 3, 4):
        if not (K._tile_legal(cin, t) if t < 4 else (cin % 128 == 0 and cout in (64, 128, 256, 512))): continue
        K._tile_cache[key] = t
        def run():
            post = K.BnBwdFuse(mask, list(zip(xs, saved)))
            K.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16", post=post, out_dtype=BF)
        res[t] = timeit(run)
    best = min(res, key=res.get)
    print('"dgrad+bn|%d|1|1|%d|%d|%d|%d|%d|1|1|1|0|1": %d,   # %s  (table had %s)' % (
        nbn, B, H, H, cin, cout, best, "  ".join("%d: %.0f us" % kv for kv in sorted(res.items())), old))
    tot_old += n * res.get(old if old in res else 2, res[best]); tot_new += n * res[best]
print("per step: table %.2f ms -> best %.2f ms" % (tot_old / 1e3, tot_new / 1e3))
