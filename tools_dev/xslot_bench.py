"""Times the fused xSlot forward / backward kernels alone (batch, slots, tokens from argv) and prints MFMA-roofline fractions.
usage: python tools_dev/xslot_bench.py [B=256] [S=300] [spc=3] [N=49] [T=3] [L=1]"""
import sys
import torch
sys.path.insert(0, '.')
from scouter_amd import kernels as K

a = [int(v) for v in sys.argv[1:]]
B, S, spc, N, T, L = (a + [256, 300, 3, 49, 3, 1][len(a):])[:6]
d = 64
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
X, PE = r(B, N, d).relu_(), r(N, d) * 0.3
tok_w, tok_b = [r(d, d) * 0.1 for _ in range(L)], [r(d) * 0.1 for _ in range(L)]
slots0 = r(S, d).abs() * 0.5
w_ih, w_hh, b_ih, b_hh = r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1
args = (X, PE, tok_w, tok_b, slots0, w_ih, w_hh, b_ih, b_hh, spc, T, 1)


import ctypes
from scouter_amd import _native


def timeit(fn, n=20):
    """kernel time from the library's own hipEvents around each launch (host / Python launch overhead excluded)"""
    fn(); fn()
    torch.cuda.synchronize()
    L = _native.lib()
    buf = ctypes.create_string_buffer(1 << 14)
    L.scouter_prof_collect(buf, len(buf))
    L.scouter_prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    L.scouter_prof_enable(0)
    L.scouter_prof_collect(buf, len(buf))
    tot = 0.0
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split("\t")[:3]
        if name.startswith("xslot"):
            tot += float(ms) / float(cnt)
    return tot * 1e-3


out = K.xslot_fwd(*args)
dlog, garea = r(B, S // spc) * 0.01, torch.full((1,), 1e-4, device='cuda')
bw = lambda: K.xslot_bwd(X, PE, tok_w, slots0, w_ih, w_hh, b_ih, b_hh, out, dlog, garea, spc, T, 1)
tfs = [timeit(lambda: K.xslot_fwd(*args)) for _ in range(8)]
tbs = [timeit(bw) for _ in range(4)]
print('fwd batches (us):', ' '.join('%.1f' % (t * 1e6) for t in tfs), '| bwd:', ' '.join('%.1f' % (t * 1e6) for t in tbs))
tf, tb = sorted(tfs)[len(tfs) // 2], sorted(tbs)[len(tbs) // 2]      # median batch of 20 launches
qk = 2.0 * S * N * d                      # one QK^T (or A.X) contraction per image
fwd_fl = B * (2.0 * L * N * d * d + T * 2 * qk + (T - 1) * 12.0 * S * d * d)
print('B=%d S=%d N=%d T=%d L=%d' % (B, S, N, T, L))
print('xslot_fwd %.1f us  %.2f TFLOP/s algorithmic = %.3f of 157.3 (QK^T+AV share %.0f%% of FLOPs)' % (
    tf * 1e6, fwd_fl / tf / 1e12, fwd_fl / tf / 157.3e12, 100 * B * T * 2 * qk / fwd_fl))
print('xslot_bwd %.1f us  (%.2fx fwd)  ~%.2f TFLOP/s on 3x fwd FLOPs' % (tb * 1e6, tb / tf, 3 * fwd_fl / tb / 1e12))
