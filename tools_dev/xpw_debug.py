"""Where does tile 5 of the fp32 fused input gradient differ from tile 2?  (development aid)"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from scouter_amd import kernels as kk
def run(B, H, W, Cin, Cout, two, with_add):
    torch.manual_seed(1)
    shape = (B, H, W, Cin)
    dy = torch.randn(B, H, W, Cout, device='cuda')
    w = torch.randn(1, 1, Cin, Cout, device='cuda') * 0.1
    add = torch.randn(*shape, device='cuda') if with_add else None
    g_, b_ = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    xs = [torch.randn(*shape, device='cuda') for _ in range(2 if two else 1)]
    outs = [kk.bn_fwd(x, g_, b_, torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'), True, True, want_mask=True) for x in xs]
    saved, mask = [o[1] for o in outs], outs[0][2]
    key = ("dgrad+bn", len(xs), with_add, False, B, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    out = {}
    for tile in (2, 5):
        kk._tile_cache[key] = tile
        post = kk.BnBwdFuse(mask, list(zip(xs, saved)))
        g = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, post=post)
        out[tile] = (g.reshape(-1, Cin), post.rows)
    g1, g4 = out[2][0], out[5][0]
    bad = ((g1 - g4).abs() > g1.abs() * 1e-4 + 1e-5)
    rows = bad.any(1).nonzero().flatten().cpu().numpy()
    cols = bad.any(0).nonzero().flatten().cpu().numpy()
    wg = out[5][1]
    print("case", (B, H, W, Cin, Cout, two, with_add), "M", g1.shape[0], "tiles", -(-g1.shape[0] // 64), "wg_per_col", wg,
          "| bad elements", int(bad.sum()), "nan", int(torch.isnan(g4).sum()), "bad rows", len(rows), "bad cols", len(cols),
          "zero-pattern mismatches", int(((g1 == 0) != (g4 == 0)).sum()))
    if len(rows):
        t = rows // 64
        print("   tiles with bad rows: first", t[:10], "count", len(np.unique(t)), "tile // wg_per_col histogram", np.bincount(np.unique(t) // wg),
              "row-in-tile histogram", np.bincount(rows % 64, minlength=64), "cols first", cols[:16], "cols % 64 hist", np.bincount(cols % 64, minlength=64))
        r = rows[0]; c = bad[r].nonzero().flatten()[:8].cpu().numpy()
        print("   row", r, "cols", c, "tile2", g1[r, c].cpu().numpy(), "tile5", g4[r, c].cpu().numpy())
for case in [(36, 28, 28, 512, 256, True, True), (36, 28, 28, 512, 256, False, True), (36, 28, 28, 512, 256, False, False), (8, 28, 28, 512, 256, True, True),
             (70, 28, 28, 512, 256, True, True), (70, 14, 14, 1024, 256, False, True), (70, 56, 56, 256, 128, True, True), (70, 56, 56, 256, 64, False, True)]:
    run(*case)
