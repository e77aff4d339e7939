"""1x1 layers of resnest26d (batch 70): fp32-MFMA kernels vs the bf16x3 plane kernels, forward / input gradient / weight
gradient -- would moving the pointwise layers onto planes pay?  usage: python tools_dev/planes_1x1_bench.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
# cin, cout, H, count
shapes = [(64, 256, 56, 3), (256, 64, 56, 1), (256, 128, 56, 1), (128, 512, 28, 2), (256, 512, 28, 1), (512, 128, 28, 1),
          (512, 256, 28, 1), (256, 1024, 14, 2), (512, 1024, 14, 1), (1024, 256, 14, 1), (1024, 512, 14, 1),
          (512, 2048, 7, 2), (1024, 2048, 7, 1), (2048, 512, 7, 1)]
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = dict(f32=0.0, f3=0.0, d32=0.0, d3=0.0, w32=0.0, w3=0.0)
print("%-22s %3s %6s | fwd fp32   x3 (tile) | dgrad fp32   x3 (tile) | wgrad fp32   x3" % ("cin,cout,H", "cnt", "GFLOP"))
for cin, cout, H, cnt in shapes:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    fl = 2.0 * B * H * H * cout * cin
    f32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True))
    xp3 = K.planes_split(x, 3)
    wf3, wd3 = K.planes_split_weight(w, 1, 3)
    tf = {t: timeit(lambda: K.conv2d_fwd_planes(xp3, wf3, 1, 1, 1, 0, 1, bn_stats=True, tile=t)) for t in K._plane_tiles(cout)}
    bt = min(tf, key=tf.get)
    y = K.conv2d_fwd(x, w, None, None, 1, 0, 1); dy = torch.randn_like(y); dyp = K.planes_split(dy, 3)
    d32 = timeit(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, 0, 1))
    td = {t: timeit(lambda: K.conv2d_dgrad_planes(dyp, wd3, tuple(x.shape), 1, 1, 1, 0, 1, tile=t)) for t in K._plane_tiles(cin)}
    bd = min(td, key=td.get)
    dw = torch.empty_like(w)
    w32 = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 1, 0, 1))
    w3 = timeit(lambda: K.conv2d_wgrad_planes(xp3, dyp, dw, 0, 1))
    print("%-22s %3d %6.2f | %7.1f %7.1f (%d)   | %7.1f %7.1f (%d)     | %7.1f %7.1f" % (
        str((cin, cout, H)), cnt, fl / 1e9, f32, tf[bt], bt, d32, td[bd], bd, w32, w3))
    for k, v in (("f32", f32), ("f3", tf[bt]), ("d32", d32), ("d3", td[bd]), ("w32", w32), ("w3", w3)):
        tot[k] += v * cnt
    tot.setdefault("best", 0.0)
    tot["best"] += cnt * (min(f32, tf[bt]) + min(d32, td[bd]) + min(w32, w3))
print("totals us/step: fwd %.0f -> %.0f   dgrad %.0f -> %.0f   wgrad %.0f -> %.0f   per-layer best of both: %.0f (fp32 all: %.0f)" % (
    tot["f32"], tot["f3"], tot["d32"], tot["d3"], tot["w32"], tot["w3"], tot["best"], tot["f32"] + tot["d32"] + tot["w32"]))
