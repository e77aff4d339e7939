"""Per-layer table of the conv kernels at a BASELINE config's shapes: every block tile for fwd / dgrad, the wgrad plan,
time x layer count, and the MFMA-time floor per layer (FLOP / 157.3 TF/s) -- shows where the convolution time goes.
usage: python tools_dev/layer_table.py [B] [arch]"""
import collections
import sys

import torch

sys.path.insert(0, '.')
from scouter_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
arch = sys.argv[2] if len(sys.argv) > 2 else 'resnest26d'
blocks = dict(resnest26d=[2, 2, 2, 2], resnest50d=[3, 4, 6, 3])[arch]
shapes = collections.OrderedDict()


def rec(cin, cout, k, s, p, g, H):
    key = (cin, cout, k, s, p, g, H)
    shapes[key] = shapes.get(key, 0) + 1


H = 112
rec(32, 32, 3, 1, 1, 1, H); rec(32, 64, 3, 1, 1, 1, H)
H = 56; inp = 64
for li, (planes, n) in enumerate(zip([64, 128, 256, 512], blocks)):
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


def timeit(fn, n=12):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


names = ['128x128', '128x64', '64x64', '128x32']
print('%-34s %3s %7s %6s | fwd us by tile %s | dgrad us by tile | wgrad us' % ('cin,cout,k,s,p,g,H', 'cnt', 'GFLOP', 'floor', names))
tot = collections.defaultdict(float)
for (cin, cout, k, s, p, g, H), cnt in shapes.items():
    x = torch.randn(B, H, H, cin, device='cuda')
    w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
    y = K.conv2d_fwd(x, w, None, None, s, p, g)
    dy = torch.randn_like(y)
    dw = torch.empty_like(w)
    fl = 2.0 * y.numel() * (cin // g) * k * k
    floor = fl / 157.3e12 * 1e6
    fk = ("fwd", False, B, H, H, cin, cout, k, k, s, p, g)
    dk = ("dgrad", False, B, H, H, cin, cout, k, k, s, p, g)
    tf, td = [], []
    for t in range(4):
        if K._tile_legal(cout // g, t):
            K._tile_cache[fk] = t
            tf.append(timeit(lambda: K.conv2d_fwd(x, w, None, None, s, p, g, bn_stats=True)))
        else:
            tf.append(float('nan'))
        if K._tile_legal(cin // g, t):
            K._tile_cache[dk] = t
            td.append(timeit(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), None, s, p, g)))
        else:
            td.append(float('nan'))
    tw = timeit(lambda: K.conv2d_wgrad(x, dy, dw, s, p, g))
    bf = min(v for v in tf if v == v); bd = min(v for v in td if v == v)
    fmt = lambda v: ' '.join('%6.1f' % q for q in v)
    print('%-34s %3d %7.2f %6.1f | %s | %s | %6.1f   best/floor f %.2f d %.2f w %.2f' % (
        str((cin, cout, k, s, p, g, H)), cnt, fl / 1e9, floor, fmt(tf), fmt(td), tw, bf / floor, bd / floor, tw / floor))
    tot['floor'] += floor * cnt; tot['fwd'] += bf * cnt; tot['dgrad'] += bd * cnt; tot['wgrad'] += tw * cnt
    for t in range(4):
        tot['f%d' % t] += (tf[t] if tf[t] == tf[t] else bf) * cnt
        tot['d%d' % t] += (td[t] if td[t] == td[t] else bd) * cnt
print('TOTAL us/pass: floor %.0f | fwd best %.0f (%.2fx) dgrad best %.0f (%.2fx) wgrad %.0f (%.2fx)' % (
    tot['floor'], tot['fwd'], tot['fwd'] / tot['floor'], tot['dgrad'], tot['dgrad'] / tot['floor'], tot['wgrad'],
    tot['wgrad'] / tot['floor']))
print('single-tile totals fwd', ['%.0f' % tot['f%d' % t] for t in range(4)], 'dgrad', ['%.0f' % tot['d%d' % t] for t in range(4)])
