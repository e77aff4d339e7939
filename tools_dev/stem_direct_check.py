"""Deep stem's first convolution at the bench's size: direct kernel (csrc/misc_ops.hip stem_direct_kernel) against the im2col +
GEMM route -- microseconds per call (hipEvents, 50 calls) and the maximum error of each against an fp64 convolution.
usage: python tools_dev/stem_direct_check.py [B H W]"""
import sys

import torch

sys.path.insert(0, ".")
import __graft_entry__ as G  # noqa: E402

G.build()
from scouter_amd import kernels as K  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (70, 224, 224)
x = torch.randn(B, 3, H, W, device="cuda")
w = torch.randn(3, 3, 3, 32, device="cuda") * 0.08
wpad = K.pad_rows(w.reshape(-1), w.numel(), 32 * 32).view(1, 1, 32, 32)


def direct():
    return K.stem_direct_fwd(x, w, bn_stats=True)[0]


def gemm():
    col = K.im2col_nchw(x, 3, 2, 1, 32)
    return K.conv2d_fwd(col, wpad, None, None, 1, 0, 1, False, True)[0]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


ref = torch.nn.functional.conv2d(x[:4].double(), w.double().permute(3, 2, 0, 1), None, 2, 1).permute(0, 2, 3, 1)
yd, yg = direct(), gemm()
byt = 4.0 * (x.numel() + yd.numel())
td, tg = timed(direct), timed(gemm)
print("direct %.1f us (%.2f TB/s of algorithmic bytes)   im2col + GEMM %.1f us   max err vs fp64: direct %.3g, GEMM %.3g"
      % (td, byt / td / 1e6, tg, float((yd[:4].double() - ref).abs().max()), float((yg[:4].double() - ref).abs().max())))
