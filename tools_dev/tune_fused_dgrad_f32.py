"""Block tiles of the fused 1x1 input gradient (conv1 + the previous block's BatchNorm-backward sums), fp32 tensors: tiles 0-3
(igemm_kernel), 4 (persistent fp32, pwp_fused_kernel), 5 (persistent bf16x3, xpw_fused_kernel) per shape of resnest26d /
resnest50d at batch B; prints the table lines.  usage: python tools_dev/tune_fused_dgrad_f32.py [B]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
tot_old = tot_new = 0.0
# (H, Cin = block input width, Cout = group width of the consuming conv1, BatchNorms finished, shortcut gradient, launches per step)
for H, cin, cout, nbn, has_add, n in [(56, 64, 64, 1, True, 1), (56, 256, 64, 2, True, 1), (56, 256, 64, 1, True, 1), (56, 256, 128, 1, True, 1), (56, 256, 128, 2, True, 1),
                                      (28, 512, 128, 2, True, 1), (28, 512, 128, 1, True, 1), (28, 512, 256, 1, True, 1), (28, 512, 256, 2, True, 1),
                                      (14, 1024, 256, 2, True, 1), (14, 1024, 256, 1, True, 1)]:
    shape = (B, H, H, cin)
    dy = torch.randn(B, H, H, cout, device='cuda')
    w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    add = torch.randn(*shape, device='cuda') if has_add else None
    xs = [torch.randn(*shape, device='cuda') for _ in range(nbn)]
    g, b = torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda')
    outs = [K.bn_fwd(x, g, b, torch.zeros(cin, device='cuda'), torch.ones(cin, device='cuda'), True, True, want_mask=True) for x in xs]
    saved, mask = [o[1] for o in outs], outs[0][2]
    key = ("dgrad+bn", nbn, has_add, False, B, H, H, cin, cout, 1, 1, 1, 0, 1)
    old = K._table_choice(key, lambda t, dry=False: True)
    res = {}
    for t in (0, 1, 2, 3, 4, 5):
        K._tile_cache[key] = t
        def run():
            post = K.BnBwdFuse(mask, list(zip(xs, saved)))
            K.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, post=post)
        try:
            res[t] = timeit(run)
        except RuntimeError:
            pass
    best = min(res, key=res.get)
    mb = B * H * H * (cout + cin * (2 + int(has_add) + nbn)) * 4 / 1e6
    print('"%s": %d,   # %s  (table had %s; %.0f MB -> %.2f TB/s)' % (K._key_str(key), best, "  ".join("%d: %.0f us" % kv for kv in sorted(res.items())), old, mb, mb / res[best]))
    tot_old += n * res.get(old, res[best]); tot_new += n * res[best]
print("sum: table %.2f ms -> best %.2f ms" % (tot_old / 1e3, tot_new / 1e3))
