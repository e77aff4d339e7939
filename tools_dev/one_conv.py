"""One convolution layer in a loop (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools_dev/one_conv.py MODE cin cout k groups H [B] [tile]     MODE = fwd | dgrad | wgrad"""
import sys, torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K
a = sys.argv[1:]
mode = a[0] if a else 'fwd'
cin, cout, k, g, H = (int(v) for v in a[1:6]) if len(a) >= 6 else (128, 256, 3, 2, 56)
B = int(a[6]) if len(a) > 6 else 70
tile = int(a[7]) if len(a) > 7 else None
p = k // 2
x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(k, k, cin // g, cout, device='cuda') * 0.05
if tile is not None:
    K._tile_cache[("fwd", False, B, H, H, cin, cout, k, k, 1, p, g)] = tile
    K._tile_cache[("dgrad", False, B, H, H, cin, cout, k, k, 1, p, g)] = tile
y = K.conv2d_fwd(x, w, None, None, 1, p, g); dy = torch.randn_like(y); dw = torch.empty_like(w)
if mode.startswith('pfwd'):                      # plane convolution: pfwd3 / pfwd1
    npl = int(mode[4:] or 3)
    xp = K.planes_split(x, npl); wf, _ = K.planes_split_weight(w, g, npl)
if mode == 'pwgrad':
    xp = K.planes_split(x, 3); dyp = K.planes_split(dy, 3)
for _ in range(10):
    if mode == 'pwgrad': K.conv2d_wgrad_planes(xp, dyp, dw, p, g)
    elif mode.startswith('pfwd'): K.conv2d_fwd_planes(xp, wf, k, k, 1, p, g, bn_stats=True, tile=tile or 0)
    elif mode == 'fwd': K.conv2d_fwd(x, w, None, None, 1, p, g, bn_stats=True)
    elif mode == 'dgrad': K.conv2d_dgrad(dy, w, tuple(x.shape), None, 1, p, g)
    else: K.conv2d_wgrad(x, dy, dw, 1, p, g)
torch.cuda.synchronize()
