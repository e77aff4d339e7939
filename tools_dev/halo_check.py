"""Resident-rows plane kernel (tile 5) against tile 4 / tile 0 at the MODEL's layer shapes, repeated: max |difference|
relative to max |y| must be fp32 rounding (1e-6); also run-to-run identity.  usage: python tools_dev/halo_check.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
torch.manual_seed(0)
for cin, cout, g, H in [(64, 128, 2, 56), (128, 256, 2, 56), (128, 256, 2, 28), (256, 512, 2, 28), (256, 512, 2, 14),
                        (512, 1024, 2, 14), (512, 1024, 2, 7)]:
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(3, 3, cin // g, cout, device='cuda') * 0.05
    xp = K.planes_split(x, 3); wf, wd = K.planes_split_weight(w, g, 3)
    ref_t = 4 if (cout // g) % 128 == 0 else 2
    y_ref, (p_ref, r_ref) = K.conv2d_fwd_planes(xp, wf, 3, 3, 1, 1, g, bn_stats=True, tile=ref_t)
    sc = float(y_ref.abs().max())
    worst, ident, sworst = 0.0, True, 0.0
    y0 = None
    for it in range(6):
        y, (p, r) = K.conv2d_fwd_planes(xp, wf, 3, 3, 1, 1, g, bn_stats=True, tile=5)
        worst = max(worst, float((y - y_ref).abs().max()) / sc)
        sworst = max(sworst, float((p.sum(0) - p_ref.sum(0)).abs().max() / p_ref.sum(0).abs().max()))
        if y0 is None: y0 = y.clone()
        else: ident = ident and bool(torch.equal(y, y0))
    if (cin // g) % 64:
        print('%-22s fwd tile5 vs tile%d: %.2e (stats %.2e) run-to-run identical: %s' % (str((cin, cout, g, H)), ref_t, worst, sworst, ident)); continue
    dy = torch.randn_like(y_ref); dyp = K.planes_split(dy, 3)
    d_ref = K.conv2d_dgrad_planes(dyp, wd, tuple(x.shape), 3, 3, 1, 1, g, tile=(4 if (cin // g) % 128 == 0 else 2))
    dworst = 0.0
    for it in range(4):
        d = K.conv2d_dgrad_planes(dyp, wd, tuple(x.shape), 3, 3, 1, 1, g, tile=5)
        dworst = max(dworst, float((d - d_ref).abs().max()) / float(d_ref.abs().max()))
    print("%-22s fwd tile5 vs tile%d: %.2e (stats %.2e) run-to-run identical: %s | dgrad %.2e" % (
        str((cin, cout, g, H)), ref_t, worst, sworst, ident, dworst))
