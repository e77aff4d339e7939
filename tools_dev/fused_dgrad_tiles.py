import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B=70
def timeit(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, cin, cout, H, ne in [("l1.b1.conv1 dgrad+2bn", 256, 64, 56, 2), ("l2.b0.conv1 dgrad+1bn", 256, 128, 56, 1), ("l2.b1.conv1 dgrad+2bn", 512, 128, 28, 2)]:
    dy = torch.randn(B, H, H, cout, device="cuda"); w = torch.randn(1, 1, cin, cout, device="cuda") * 0.05
    addend = torch.randn(B, H, H, cin, device="cuda")
    xs = [torch.randn(B, H, H, cin, device="cuda") for _ in range(ne)]
    gam, bet = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    saveds = []; mask = None
    for i, x in enumerate(xs):
        rm, rv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
        if i == 0: _, sv, mask = K.bn_fwd(x, gam, bet, rm, rv, True, True, want_mask=True)
        else: _, sv = K.bn_fwd(x, gam, bet, rm, rv, True, False)
        saveds.append(sv)
    shape = (B, H, H, cin)
    res = []
    for t in range(4):
        if not K._tile_legal(cin, t): res.append("  n/a"); continue
        K._tile_cache[("dgrad", False, B, H, H, cin, cout, 1, 1, 1, 0, 1)] = t
        K._tile_cache[("dgrad+bn", ne, True, False, B, H, H, cin, cout, 1, 1, 1, 0, 1)] = t
        tp = timeit(lambda: K.conv2d_dgrad(dy, w, shape, addend, 1, 0, 1))
        tf = timeit(lambda: K.conv2d_dgrad(dy, w, shape, addend, 1, 0, 1, post=K.BnBwdFuse(mask, list(zip(xs, saveds)))))
        res.append("%d: %.0f/%.0f" % (t, tp, tf))
    mb_p = (B*H*H*(cout + 2*cin))*4/1e6; mb_f = mb_p + B*H*H*cin*4*ne/1e6
    print(name, "plain/fused us by tile (0=128x128 1=128x64 2=64x64 3=128x32):", res, "| MB plain %.0f fused %.0f" % (mb_p, mb_f))
# pure streaming reference: bn_bwd_apply-like (reads 2, writes 1) and a 4-read 1-write torch expression
x = torch.randn(B,56,56,256, device="cuda"); y = torch.randn_like(x); z = torch.randn_like(x); u = torch.randn_like(x)
out = torch.empty_like(x)
t = timeit(lambda: torch.add(x, y, out=out)); print("torch add (2r+1w, 674 MB): %.0f us = %.2f TB/s" % (t, 674.3/t*1e-0/1e3*1e3/1e3))
