"""can HBM-bound BatchNorm apply passes run UNDER the MFMA-bound convolutions of the other half batch?

A chain of L x [1x1 convolution C -> C with fused BatchNorm statistics, finalize over ALL images, apply + ReLU] at the
benchmark's map sizes, real kernels through the C ABI, three schedules:
  full      one stream, the whole batch per launch (what the step does today)
  halves    two streams, images [0, B/2) and [B/2, B); joint statistics (both halves' partial rows feed ONE finalize);
            no ordering between the halves beyond the data dependencies (the hardware picks)
  stagger   the same, but apply(second half) waits for apply(first half): conv_{l+1}(first) runs next to apply_l(second)
plus the pair experiment: conv(first half) on one stream next to apply(second half) on the other, against each alone.

usage: python tools_dev/overlap_probe.py [B] [H] [C] [L]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from scouter_amd import _native, kernels as K   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
H = int(sys.argv[2]) if len(sys.argv) > 2 else 56
C = int(sys.argv[3]) if len(sys.argv) > 3 else 128
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 8
TILE = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = torch.device("cuda:0")
L = _native.lib()
torch.manual_seed(0)
x0 = torch.randn(B, H, H, C, device=dev)
ws_ = [torch.randn(1, 1, C, C, device=dev) * (1.0 / C ** 0.5) for _ in range(NL)]
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
raw = [torch.empty(B, H, H, C, device=dev) for _ in range(NL)]
act = [torch.empty(B, H, H, C, device=dev) for _ in range(NL)]
saved = [torch.empty(4, C, device=dev) for _ in range(NL)]
Bh = B // 2
rows_full = L.scouter_conv2d_fwd_bn_partial_rows(B, H, H, C, C, 1, 1, 1, 0, 1, TILE)
rows_a = L.scouter_conv2d_fwd_bn_partial_rows(Bh, H, H, C, C, 1, 1, 1, 0, 1, TILE)
rows_b = L.scouter_conv2d_fwd_bn_partial_rows(B - Bh, H, H, C, C, 1, 1, 1, 0, 1, TILE)
part = [torch.empty(rows_a + rows_b + rows_full, C, 2, dtype=torch.float64, device=dev) for _ in range(NL)]
wsb = torch.empty(L.scouter_colreduce_workspace_bytes(B * H * H, C) + 8 * C + 64, dtype=torch.uint8, device=dev)
img = H * H * C * 4      # bytes per image


def P(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off)


def conv(x, w, y, prt, prt_off_rows, b0, nb, st):
    _native.check(L.scouter_conv2d_fwd_f32(P(x, b0 * img), P(w), None, None, P(y, b0 * img), P(prt, prt_off_rows * C * 16),
                                           nb, H, H, C, C, 1, 1, 1, 0, 1, 0, TILE, st), "conv")


def finalize(x, prt, nrows, sv, st):
    _native.check(L.scouter_bn_fwd_f32(P(x), None, None, B * H * H, C, P(gam), P(bet), P(rm), P(rv), 0.1, 1e-5, 1, 0,
                                       P(sv), P(sv, 4 * C), P(sv, 8 * C), P(sv, 12 * C), P(prt), nrows, None, None, 0, None,
                                       P(wsb), wsb.numel(), st), "finalize")


def apply(x, sv, y, b0, nb, st):
    _native.check(L.scouter_bn_apply_f32(P(x, b0 * img), P(sv), P(y, b0 * img), nb * H * H, C, 1, st), "apply")


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run_full():
    st = s1.cuda_stream
    x = x0
    for l in range(NL):
        conv(x, ws_[l], raw[l], part[l], 0, 0, B, st)
        finalize(raw[l], part[l], rows_full, saved[l], st)
        apply(raw[l], saved[l], act[l], 0, B, st)
        x = act[l]


def run_halves(stagger):
    a, b = s1.cuda_stream, s2.cuda_stream
    x = x0
    for l in range(NL):
        conv(x, ws_[l], raw[l], part[l], 0, 0, Bh, a)
        conv(x, ws_[l], raw[l], part[l], rows_a, Bh, B - Bh, b)
        e = torch.cuda.Event(); e.record(s2); s1.wait_event(e)
        finalize(raw[l], part[l], rows_a + rows_b, saved[l], a)
        apply(raw[l], saved[l], act[l], 0, Bh, a)
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)      # (finalize [+ apply(A) when staggered] done)
        if not stagger:
            pass
        apply(raw[l], saved[l], act[l], Bh, B - Bh, b)
        x = act[l]
    s1.wait_stream(s2)


def run_halves_greedy():
    a, b = s1.cuda_stream, s2.cuda_stream
    x = x0
    for l in range(NL):
        conv(x, ws_[l], raw[l], part[l], 0, 0, Bh, a)
        conv(x, ws_[l], raw[l], part[l], rows_a, Bh, B - Bh, b)
        e = torch.cuda.Event(); e.record(s2); s1.wait_event(e)
        finalize(raw[l], part[l], rows_a + rows_b, saved[l], a)
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)
        apply(raw[l], saved[l], act[l], 0, Bh, a)
        apply(raw[l], saved[l], act[l], Bh, B - Bh, b)
        x = act[l]
    s1.wait_stream(s2)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6 / NL        # us per layer


def pair():
    """conv(first half) || apply(second half): each alone, back to back on one stream, and on two streams."""
    a, b = s1.cuda_stream, s2.cuda_stream
    finalize(raw[0], part[0], rows_full, saved[0], a)
    torch.cuda.synchronize()

    def t(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    tc = t(lambda: conv(x0, ws_[0], raw[1], part[1], 0, 0, Bh, a))
    ta = t(lambda: apply(raw[0], saved[0], act[0], Bh, B - Bh, a))

    def serial():
        conv(x0, ws_[0], raw[1], part[1], 0, 0, Bh, a)
        apply(raw[0], saved[0], act[0], Bh, B - Bh, a)

    def both():
        conv(x0, ws_[0], raw[1], part[1], 0, 0, Bh, a)
        apply(raw[0], saved[0], act[0], Bh, B - Bh, b)
    return tc, ta, t(serial), t(both)


run_full(); torch.cuda.synchronize()
ref = act[-1].clone()
run_halves(True); torch.cuda.synchronize()
print("max |halves - full| on the last activation: %.3g (statistics summed in another order)"
      % float((act[-1] - ref).abs().max()))
flops = 2.0 * B * H * H * C * C
byts_conv, byts_apply = 8.0 * B * H * H * C, 8.0 * B * H * H * C
res = {}
for name, fn in (("full", run_full), ("halves_greedy", run_halves_greedy), ("halves_stagger", lambda: run_halves(True)),
                 ("full", run_full), ("halves_greedy", run_halves_greedy), ("halves_stagger", lambda: run_halves(True))):
    res.setdefault(name, []).append(timeit(fn))
print("B=%d H=%d C=%d layers=%d tile=%d: conv %.1f GF / %.0f MB, apply %.0f MB per layer" %
      (B, H, C, NL, TILE, flops / 1e9, byts_conv / 1e6, byts_apply / 1e6))
for k, v in res.items():
    print("  %-16s %s us/layer" % (k, " ".join("%.1f" % u for u in v)))
tc, ta, ts, tb = pair()
print("pair: conv(half) alone %.1f us, apply(half) alone %.1f us, serial %.1f us, two streams %.1f us -> hidden %.0f %% "
      "of the shorter one" % (tc, ta, ts, tb, 100.0 * (ts - tb) / min(tc, ta)))
