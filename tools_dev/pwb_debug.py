"""Where does tile 4 of the typed fused input gradient differ from tile 1?  (development aid)"""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from scouter_amd import kernels as kk
BF16 = torch.bfloat16
def run(B, H, W, Cin, Cout, two, with_add, use_mask=True):
    torch.manual_seed(1)
    shape = (B, H, W, Cin)
    dy = torch.randn(B, H, W, Cout, device='cuda').to(BF16)
    w = torch.randn(1, 1, Cin, Cout, device='cuda') * 0.1
    add = torch.randn(*shape, device='cuda').to(BF16) if with_add else None
    g_, b_ = torch.ones(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    xs = [torch.randn(*shape, device='cuda').to(BF16) for _ in range(2 if two else 1)]
    saved = []
    mask = None
    for i, xh in enumerate(xs):
        rm, rv = torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda')
        if i == 0 and use_mask:
            _, sv, mask = kk.bn_fwd(xh.float(), g_, b_, rm, rv, True, True, want_mask=True)
        else:
            sv = kk.bn_stats(xh.float(), g_, b_, rm, rv, True)
        saved.append(sv)
    key = ("dgrad+bn", len(xs), with_add, True, B, H, W, Cin, Cout, 1, 1, 1, 0, 1)
    out = {}
    for tile in (1, 4):
        kk._tile_cache[key] = tile
        post = kk.BnBwdFuse(mask, list(zip(xs, saved)))
        g = kk.conv2d_dgrad(dy, w, shape, add, 1, 0, 1, precision="bf16", post=post, out_dtype=BF16)
        out[tile] = (g.float().reshape(-1, Cin), post.rows)
    g1, g4 = out[1][0], out[4][0]
    bad = ((g1 - g4).abs() > g1.abs() * 2.0 ** -7 + 2e-6)
    rows = bad.any(1).nonzero().flatten().cpu().numpy()
    cols = bad.any(0).nonzero().flatten().cpu().numpy()
    wg = out[4][1]
    print("case", (B, H, W, Cin, Cout, two, with_add, use_mask), "M", g1.shape[0], "tiles", -(-g1.shape[0] // 64), "wg_per_col", wg,
          "| bad elements", int(bad.sum()), "nan", int(torch.isnan(g4).sum()), "bad rows", len(rows), "bad cols", len(cols))
    if len(rows):
        t = rows // 64
        print("   tiles with bad rows: first", t[:10], "count", len(np.unique(t)), "tile // wg_per_col histogram", np.bincount(np.unique(t) // wg),
              "row-in-tile histogram", np.bincount(rows % 64, minlength=64), "cols first", cols[:16])
        r = rows[0]; c = bad[r].nonzero().flatten()[:8].cpu().numpy()
        print("   row", r, "cols", c, "tile1", g1[r, c].cpu().numpy(), "tile4", g4[r, c].cpu().numpy())
for case in [(40, 56, 56, 256, 64, False, True), (40, 56, 56, 256, 64, False, False), (40, 56, 56, 256, 64, False, True, False), (10, 56, 56, 256, 64, False, True),
             (20, 56, 56, 256, 64, False, True), (40, 28, 28, 512, 128, False, True), (4, 56, 56, 256, 64, False, True), (4, 56, 56, 256, 64, True, True),
             (4, 28, 28, 512, 128, False, True), (8, 14, 14, 1024, 256, False, True), (32, 7, 7, 2048, 512, False, True), (32, 7, 7, 2048, 512, True, True)]:
    run(*case)
