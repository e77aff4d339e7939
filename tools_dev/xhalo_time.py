"""Timing only (no checks) of csrc/conv_xhalo.hip at the benchmark's shapes -- for the XH_ABLATE builds of tools_dev/xhalo_ablate.sh
(SCOUTER_HIP_LIB=build_dev/libscouter_xh<N>.so)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scouter_amd import kernels as K


def timeit(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 70
out = []
x = torch.randn(B, 112, 112, 32, device="cuda"); w = torch.randn(3, 3, 32, 32, device="cuda") * 0.05
wf, wd = K.planes_split_weight(w, 1, 3)
out.append("fwd 112^2 32->32 %.1f" % timeit(lambda: K.conv2d_fwd_x3(x, wf, None, False, True, tile=7, kh=3, pad=1, groups=1)))
out.append("dgrad %.1f" % timeit(lambda: K.conv2d_dgrad_x3(x, wd, tuple(x.shape), None, tile=7, kh=3, pad=1, groups=1)))
dy = torch.randn(B, 112, 112, 64, device="cuda"); w2 = torch.randn(3, 3, 32, 64, device="cuda") * 0.05
_, wd2 = K.planes_split_weight(w2, 1, 3, fwd=False, dgrad=True)
out.append("dgrad 112^2 K=64 %.1f" % timeit(lambda: K.conv2d_dgrad_x3(dy, wd2, (B, 112, 112, 32), None, tile=7, kh=3, pad=1, groups=1)))
dy3 = torch.randn(B, 56, 56, 128, device="cuda"); w3 = torch.randn(3, 3, 32, 128, device="cuda") * 0.05
_, wd3 = K.planes_split_weight(w3, 2, 3, fwd=False, dgrad=True)
out.append("dgrad 56^2 g2 K=64 %.1f" % timeit(lambda: K.conv2d_dgrad_x3(dy3, wd3, (B, 56, 56, 64), None, tile=7, kh=3, pad=1, groups=2)))
print(os.environ.get("SCOUTER_HIP_LIB", "product"), "|", " | ".join(out), flush=True)
