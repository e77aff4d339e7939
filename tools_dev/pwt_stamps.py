"""Where a chunk of the tap-fused plane weight gradient spends its cycles (wave 0 of every workgroup, -DPWT_STAMPS build):
DMA wait + barrier | chunk prologue (masks, DMA issue, first fragment reads) | the 18 MFMA groups.  108 MFMAs x 33 cycles =
3 564 cycles per chunk is the floor.  usage: SCOUTER_HIP_LIB=build_dev/libscouter_pwt.so python tools_dev/pwt_stamps.py [plan]"""
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K, _native
B, plan = 70, int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = _native.lib()
for cin, cout, g, H in [(128, 256, 2, 56), (256, 512, 2, 28), (512, 1024, 2, 14), (512, 1024, 2, 7)]:
    x = torch.randn(B, H, H, cin, device='cuda'); dy = torch.randn(B, H, H, cout, device='cuda')
    xp, dyp = K.planes_split(x, 3), K.planes_split(dy, 3)
    dw = torch.empty(3, 3, cin // g, cout, device='cuda')
    key = ("pwgrad", 3, B, H, H, cin, cout, 3, 3, 1, g)
    K._tile_cache[key] = plan
    for _ in range(3):
        K.conv2d_wgrad_planes(xp, dyp, dw, 1, g)
    torch.cuda.synchronize()
    nbytes = L.scouter_conv2d_wgrad_planes_workspace_bytes(B, H, H, cin, cout, 3, 3, g, plan)
    ws = K.workspace(nbytes, x.device)
    tiles = (cin // g // 64) * (cout // g // 64) * g
    M = B * H * H
    chunks = -(-M // 32)
    # splits as the library plans them: budget 256 << (plan & 3) workgroups
    want = max(1, (256 << (plan & 3)) // tiles)
    cps = max(8, -(-chunks // want)); splits = -(-M // (cps * 32))
    nblk = tiles * splits
    st = ws[nbytes - nblk * 48:nbytes].view(torch.int64).view(nblk, 6).cpu().double()
    kt = cps
    print("%s plan %d: %d workgroups x %d chunks; per chunk: wait+barrier %.0f, prologue %.0f, MFMA groups %.0f cycles (floor 3564); "
          "whole workgroup %.0f cycles = %.1f us at 2.4 GHz" % ((cin, cout, g, H), plan, nblk, kt, st[:, 0].mean() / kt,
          st[:, 1].mean() / kt, st[:, 2].mean() / kt, st[:, 3].mean(), st[:, 3].mean() / 2400))
    ghz = (st[:, 3] / (st[:, 4] / 100e6)).mean() / 1e9
    span = (st[:, 5] + st[:, 4]).max() - st[:, 5].min()
    print("    shader clock DURING the kernel (s_memtime cycles / s_memrealtime, per workgroup): %.3f GHz; first start -> last end "
          "%.1f us (100 MHz counter)" % (ghz, span / 100.0))
