import sys, torch, numpy as np
sys.path.insert(0, '.')
import torch.nn.functional as F
from scouter_amd import kernels as K
torch.manual_seed(0)
for (B, H, Cin, Cout, k, g) in [(4, 28, 256, 512, 3, 2), (4, 28, 256, 512, 1, 1), (2, 56, 128, 256, 3, 2)]:
    for mean in (0.0, 0.5):
        x = (torch.randn(B, H, H, Cin) + mean).relu().cuda() if mean else torch.randn(B, H, H, Cin).cuda()
        w = (torch.randn(k, k, Cin // g, Cout) * 0.05).cuda()
        p = k // 2
        ref = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.permute(3, 2, 0, 1).cpu().double(), None, 1, p, 1, g).permute(0, 2, 3, 1)
        y32 = K.conv2d_fwd(x, w, None, None, 1, p, g).cpu().double()
        y3 = K.conv2d_fwd_planes(K.planes_split(x, 3), K.planes_split_weight(w, g, 3)[0], k, k, 1, p, g).cpu().double()
        sc = ref.abs().mean()
        for name, y in (("fp32-mfma", y32), ("bf16x3", y3)):
            e = y - ref
            print("%-28s mean %.1f %-10s: mean err %+.3e  rms %.3e  max %.3e   (rel to mean|y| %.3g: bias %+.2e rms %.2e)" % (
                str((B, H, Cin, Cout, k, g)), mean, name, e.mean(), e.pow(2).mean().sqrt(), e.abs().max(), sc, e.mean() / sc, e.pow(2).mean().sqrt() / sc))
