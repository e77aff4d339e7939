"""Per-layer relative deviation of the bf16 conv kernels from the fp32 ones on the resnest26d shapes (should be ~3e-3)."""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
shapes = [(32, 32, 3, 1, 1, 1, 112), (32, 64, 3, 1, 1, 1, 112), (64, 64, 1, 1, 0, 1, 56), (64, 128, 3, 1, 1, 2, 56), (64, 256, 1, 1, 0, 1, 56),
          (256, 64, 1, 1, 0, 1, 56), (256, 128, 1, 1, 0, 1, 56), (128, 256, 3, 1, 1, 2, 56), (128, 512, 1, 1, 0, 1, 28), (512, 128, 1, 1, 0, 1, 28),
          (256, 512, 3, 1, 1, 2, 28), (512, 1024, 3, 1, 1, 2, 14), (1024, 2048, 1, 1, 0, 1, 7), (2048, 512, 1, 1, 0, 1, 7), (512, 1024, 3, 1, 1, 2, 7), (2048, 64, 1, 1, 0, 1, 7)]
def rel(a, b): return float((a - b).double().pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())
for tile in (None, 0, 1, 2, 3):
    print("tile", tile)
    for (cin, cout, k, s, p, g, H) in shapes:
        x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
        out = {}
        for prec in ("fp32", "bf16"):
            K.PRECISION = prec
            K._tile_cache.clear()
            if tile is not None:
                K.AUTOTUNE = False
                orig = K._pick_tile
                K._pick_tile = lambda key, launch, candidates=(0, 1, 2, 3), t=tile: (t if launch(t, dry=True) else -1) if key[0] != "wgrad" else -1
            y = K.conv2d_fwd(x, w, None, None, s, p, g)
            dy = torch.randn_like(y)
            dx = K.conv2d_dgrad(dy, w, tuple(x.shape), None, s, p, g)
            dw = torch.empty_like(w); K.conv2d_wgrad(x, dy, dw, s, p, g)
            if tile is not None: K._pick_tile = orig; K.AUTOTUNE = True
            torch.manual_seed(0)
            out[prec] = (y, dx, dw)
        torch.manual_seed(1)
        # dy differs between precisions (fresh randn): recompute the bf16 ones with the fp32 dy for a fair comparison
        K.PRECISION = "fp32"
        print("  %-32s fwd %.2e" % (str((cin, cout, k, s, p, g, H)), rel(out["bf16"][0], out["fp32"][0])))
K.PRECISION = "fp32"
