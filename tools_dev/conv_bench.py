"""Per-layer timing of the conv kernels for every conv shape of resnest26d @224, batch 70 (fwd / dgrad / wgrad)."""
import sys, time, collections
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
from scouter_amd.timm import create_model
from scouter_amd.nn_hip import Conv2d, StemConv2d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"   # "bf16": matrix inputs rounded to bf16
m = create_model('resnest26d', num_classes=10)
shapes = collections.OrderedDict()
# walk the net with shape tracking: replicate spatial sizes
def rec(cin, cout, k, s, p, g, H):
    key = (cin, cout, k, s, p, g, H)
    shapes[key] = shapes.get(key, 0) + 1
H = 112
rec(32, 32, 3, 1, 1, 1, H); rec(32, 64, 3, 1, 1, 1, H)
H = 56; inp = 64
for li, (planes, n) in enumerate(zip([64, 128, 256, 512], [2, 2, 2, 2])):
    for bi in range(n):
        stride = 2 if (li > 0 and bi == 0) else 1
        rec(inp, planes, 1, 1, 0, 1, H)
        rec(planes, 2 * planes, 3, 1, 1, 2, H)
        Ho = H // stride
        rec(planes, 4 * planes, 1, 1, 0, 1, Ho)
        if bi == 0:
            rec(inp, 4 * planes, 1, 1, 0, 1, Ho)
        inp = 4 * planes
        H = Ho
rec(2048, 64, 1, 1, 0, 1, 7)
tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0); totf = 0.0
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
print('%-40s %5s %9s | %8s %6s | %8s %6s | %8s %6s' % ('cin,cout,k,s,p,g,H', 'cnt', 'GFLOP', 'fwd us', 'TF', 'dgrad us', 'TF', 'wgrad us', 'TF'))
for (cin, cout, k, s, p, g, H), cnt in shapes.items():
    x = torch.randn(B, H, H, cin, device='cuda')
    w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    y = K.conv2d_fwd(x, w, None, None, s, p, g, precision=PREC)
    dy = torch.randn_like(y)
    dw = torch.empty_like(w)
    fl = 2.0 * y.numel() * (cin // g) * k * k
    tf = timeit(lambda: K.conv2d_fwd(x, w, None, None, s, p, g, precision=PREC))
    td = timeit(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), None, s, p, g, precision=PREC))
    tw = timeit(lambda: K.conv2d_wgrad(x, dy, dw, s, p, g, precision=PREC))
    print('%-40s %5d %9.2f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f' % (str((cin, cout, k, s, p, g, H)), cnt, fl / 1e9, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12))
    tot['fwd'] += tf * cnt; tot['dgrad'] += td * cnt; tot['wgrad'] += tw * cnt; totf += fl * cnt
print('TOTAL ms  fwd %.2f dgrad %.2f wgrad %.2f | GFLOP/pass %.1f  => TF fwd %.1f dgrad %.1f wgrad %.1f' % (tot['fwd'] * 1e3, tot['dgrad'] * 1e3, tot['wgrad'] * 1e3, totf / 1e9, totf / tot['fwd'] / 1e12, totf / tot['dgrad'] / 1e12, totf / tot['wgrad'] / 1e12))
