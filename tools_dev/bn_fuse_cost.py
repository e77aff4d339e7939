"""Per layer of the benchmark step: what fusing the BatchNorm-backward reductions into the producing input-gradient
epilogue costs that kernel, next to what it saves in the BatchNorm backward (reduce pass + masked-gradient copy).
usage: python tools_dev/bn_fuse_cost.py [B]"""
import sys

import torch

sys.path.insert(0, '.')
from scouter_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70


def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (name, Cin (dx channels), Cout (dy channels), k, pad, groups, H, entries, addend, planes)
L = []
H, inp = 56, 64
for li, planes in enumerate([64, 128, 256, 512]):
    for bi in range(2):
        stride = 2 if (li > 0 and bi == 0) else 1
        if not (li == 0 and bi == 0):
            # conv1's dgrad produces the previous block's output gradient: bn3 (+ downsample bn if that block had one)
            L.append(("l%d.b%d.conv1->prev bn3%s" % (li + 1, bi, "+ds" if bi == 1 or True and bi == 0 and False else ""),
                      inp, planes, 1, 0, 1, H, 2 if bi == 1 else 1, True, False))
        L.append(("l%d.b%d.conv2->bn1" % (li + 1, bi), planes, 2 * planes, 3, 1, 2, H, 1, False, True))
        H //= stride
        inp = 4 * planes
L.append(("head conv1x1->l4.b1 bn3", 2048, 64, 1, 0, 1, 7, 1, False, False))
L.append(("stem conv3->bn", 32, 64, 3, 1, 1, 112, 1, False, False))
L.append(("stem conv2->bn", 32, 32, 3, 1, 1, 112, 1, False, False))

print("%-28s %22s | dgrad plain  fused   d | bn_bwd plain  ext   d | net us" % ("layer", "Cin,Cout,k,g,H,entries"))
tot = 0.0
for name, cin, cout, k, pad, g, H, ne, add, planes in L:
    dy = torch.randn(B, H, H, cout, device="cuda")
    w = torch.randn(k, k, cin // g, cout, device="cuda") * 0.05
    addend = torch.randn(B, H, H, cin, device="cuda") if add else None
    xs = [torch.randn(B, H, H, cin, device="cuda") for _ in range(ne)]
    gam, bet = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    saveds, mask = [], None
    for i, x in enumerate(xs):
        rm, rv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
        if i == 0:
            _, sv, mask = K.bn_fwd(x, gam, bet, rm, rv, True, True, want_mask=True)
        else:
            _, sv = K.bn_fwd(x, gam, bet, rm, rv, True, False)
        saveds.append(sv)
    shape = (B, H, H, cin)
    if planes and (cin // g) % 64 == 0:
        dyp = K.planes_split(dy, 3)
        _, wd = K.planes_split_weight(w, g, 3)
        run = lambda post: K.conv2d_dgrad_planes(dyp, wd, shape, k, k, 1, pad, g, addend, post=post)
    else:
        run = lambda post: K.conv2d_dgrad(dy, w, shape, addend, 1, pad, g, post=post)
    mk = lambda: K.BnBwdFuse(mask, list(zip(xs, saveds)))
    d0 = run(None)
    post = mk(); gf = run(post)
    t_plain = timeit(lambda: run(None))
    t_fused = timeit(lambda: run(mk()))
    dg, db = torch.zeros(cin, device="cuda"), torch.zeros(cin, device="cuda")

    def bn_plain():
        _, gout = K.bn_bwd(d0, None, xs[0], saveds[0], True, dg, db, True, mask=mask)
        if ne > 1:
            K.bn_bwd(gout, None, xs[1], saveds[1], True, dg, db)

    def bn_ext():
        K.bn_bwd(gf, None, xs[0], saveds[0], True, dg, db, True, ext=post.ext(0))
        if ne > 1:
            K.bn_bwd(gf, None, xs[1], saveds[1], True, dg, db, ext=post.ext(1))
    b_plain, b_ext = timeit(bn_plain), timeit(bn_ext)
    net = (t_fused - t_plain) + (b_ext - b_plain)
    tot += net
    print("%-28s %22s | %8.1f %8.1f %+6.1f | %8.1f %8.1f %+6.1f | %+7.1f" % (
        name, "%d,%d,%d,%d,%d,%d" % (cin, cout, k, g, H, ne), t_plain, t_fused, t_fused - t_plain, b_plain, b_ext,
        b_ext - b_plain, net))
print("net over the listed layers: %+.1f us per step" % tot)
