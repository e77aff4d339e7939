"""Deep 1x1 layers of resnest26d / resnest50d: exact-fp32 MFMA kernels (the static table's tile) vs the register-split
bf16x3 GEMM (csrc/conv_x3.hip) per block tile -- forward (+ fused BatchNorm statistics), plain input gradient, input
gradient with the fused BatchNorm-backward epilogue.  usage: python tools_dev/x3_bench.py [B] [arch]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
# cin, cout, H, count (forward / input gradient launches per step), fused: conv1 of a block (input gradient carries the epilogue)
shapes = [(64, 256, 56, 3, 0), (256, 64, 56, 1, 1), (256, 128, 56, 1, 1), (128, 512, 28, 2, 0), (256, 512, 28, 1, 0),
          (512, 128, 28, 1, 1), (512, 256, 28, 1, 1), (256, 1024, 14, 2, 0), (512, 1024, 14, 1, 0), (1024, 256, 14, 1, 1),
          (1024, 512, 14, 1, 1), (512, 2048, 7, 2, 0), (1024, 2048, 7, 1, 0), (2048, 512, 7, 1, 1)]
def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = dict(f32=0.0, fx=0.0, d32=0.0, dx=0.0, u32=0.0, ux=0.0)
print("%-20s %5s | fwd fp32 | x3 by tile 0-3            | dgrad fp32 | x3 by tile                | fused fp32 | x3 by tile" % ("cin,cout,H", "GF"))
for cin, cout, H, cnt, fused in shapes:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    fl = 2.0 * B * H * H * cout * cin
    wf, wd = K.planes_split_weight(w, 1, 3)
    f32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True))
    fx = [timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t)) if K._x3_tile_ok(t, cout) else float('nan') for t in range(4)]
    dy = torch.randn(B, H, H, cout, device='cuda'); add = torch.randn(B, H, H, cin, device='cuda')
    xs = tuple(x.shape)
    d32 = timeit(lambda: K.conv2d_dgrad(dy, w, xs, add, 1, 0, 1))
    dx = [timeit(lambda: K.conv2d_dgrad_x3(dy, wd, xs, addend=add, tile=t)) if K._x3_tile_ok(t, cin) else float('nan') for t in range(4)]
    g_, b_ = torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda')
    y1, saved, mask = K.bn_fwd(x, g_, b_, torch.zeros(cin, device='cuda'), torch.ones(cin, device='cuda'), True, True, want_mask=True)
    def fz32():
        post = K.BnBwdFuse(mask, [(x, saved)]); K.conv2d_dgrad(dy, w, xs, add, 1, 0, 1, post=post)
    def fzx(t):
        post = K.BnBwdFuse(mask, [(x, saved)]); K.conv2d_dgrad_x3(dy, wd, xs, addend=add, post=post, tile=t)
    u32 = timeit(fz32)
    ux = [timeit(lambda: fzx(t)) if K._x3_tile_ok(t, cin) else float('nan') for t in range(4)]
    fm = lambda v: ' '.join('%6.1f' % q for q in v)
    print("%-20s %5.1f | %8.1f | %s | %10.1f | %s | %10.1f | %s" % (str((cin, cout, H)), fl / 1e9, f32, fm(fx), d32, fm(dx), u32, fm(ux)))
    b = lambda v: min(q for q in v if q == q)
    tot["f32"] += f32 * cnt; tot["fx"] += min(f32, b(fx)) * cnt
    if fused:
        tot["u32"] += u32 * cnt; tot["ux"] += min(u32, b(ux)) * cnt
    else:
        tot["d32"] += d32 * cnt; tot["dx"] += min(d32, b(dx)) * cnt
print("per step, best of both per layer: fwd %.0f -> %.0f us, plain dgrad %.0f -> %.0f, fused dgrad %.0f -> %.0f" % (
    tot["f32"], tot["fx"], tot["d32"], tot["dx"], tot["u32"], tot["ux"]))
