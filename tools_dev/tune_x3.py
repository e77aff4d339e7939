"""Adds the block-tile entries of the register-split bf16x3 GEMM (csrc/conv_x3.hip) to the static table: for every 1x1
forward / plain input-gradient key the table already holds (all BASELINE configs, their 260x260 variants, batch 8 / 35) whose
layer qualifies (kernels.x3_eligible), tiles 0-3 are timed -- forward with the fused BatchNorm statistics, the input gradient
with an addend, as the model launches them -- 10 launches per candidate, best of two batches.  Every tile gives the same bits,
so the choice is free.  Prints the fp32 kernel's time (the table's tile) next to it.
usage (GPU box): python tools_dev/tune_x3.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scouter_amd import kernels as K   # noqa: E402

path = os.path.join(ROOT, "scouter_amd", "tuning", "gfx950.json")
out = sys.argv[1] if len(sys.argv) > 1 else path
doc = json.load(open(path))
ch = doc["choices"]


def timeit(fn, n=10):
    fn(); fn()
    best = None
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / n * 1e3
        best = t if best is None else min(best, t)
    return best


tot32 = totx = 0.0
for ks in sorted(k for k in list(ch) if k.startswith(("fwd|0|", "dgrad|0|"))):
    p = ks.split("|")
    mode = p[0]
    B, H, W, Cin, Cout, kh, kw, stride, pad, g = [int(v) for v in p[2:]]
    if not K.x3_eligible(Cin, Cout, kh, kw, stride, pad, g, False) or B * H * W < 1024:
        continue
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(1, 1, Cin, Cout, device="cuda") * 0.05
    dy = torch.randn(B, H, W, Cout, device="cuda"); add = torch.randn(B, H, W, Cin, device="cuda")
    wf, wd = K.planes_split_weight(w, 1, 3)
    N = Cout if mode == "fwd" else Cin
    res = {}
    for t in (0, 1, 2, 3):
        if not K._x3_tile_ok(t, N):
            continue
        if mode == "fwd":
            res[t] = timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t))
        else:
            res[t] = timeit(lambda: K.conv2d_dgrad_x3(dy, wd, (B, H, W, Cin), addend=add, tile=t))
    if mode == "fwd":
        t32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True))
    else:
        t32 = timeit(lambda: K.conv2d_dgrad(dy, w, (B, H, W, Cin), add, 1, 0, 1))
    best = min(res, key=res.get)
    ch["|".join(["x" + mode, "3"] + p[2:7])] = best
    tot32 += t32; totx += res[best]
    print("%-44s fp32 %7.1f us | x3 %s -> tile %d %7.1f us (%.2fx)" % (
        ks, t32, " ".join("%d:%6.1f" % kv for kv in sorted(res.items())), best, res[best], t32 / res[best]), flush=True)
print("sum over the table's shapes: fp32 %.0f us, x3 %.0f us" % (tot32, totx))
doc["choices"] = dict(sorted(ch.items()))
with open(out, "w") as f:
    json.dump(doc, f, indent=0)
print("wrote", out)

# ---- the 3x3 layers with 32-channel groups (kernels.x3_conv_eligible): forward and the (fused) input gradient
tot32 = totx = 0.0
for ks in sorted(k for k in list(ch) if k.startswith(("fwd|0|", "dgrad+bn|"))):
    p = ks.split("|")
    mode = p[0]
    if mode == "fwd":
        B, H, W, Cin, Cout, kh, kw, stride, pad, g = [int(v) for v in p[2:]]
        nent, has_add = 0, False
    else:
        nent, has_add = int(p[1]), p[2] == "1"
        if p[3] != "0":
            continue
        B, H, W, Cin, Cout, kh, kw, stride, pad, g = [int(v) for v in p[4:]]
    pw = mode != "fwd" and K.x3_eligible(Cin, Cout, kh, kw, stride, pad, g, False) and Cout >= K.X3_FUSED_MIN_K
    if not (K.x3_conv_eligible(Cin, Cout, kh, kw, stride, pad, g, False) and mode == "fwd") and not pw:
        continue
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(kh, kh, Cin // g, Cout, device="cuda") * 0.05
    dy = torch.randn(B, H, W, Cout, device="cuda")
    add = torch.randn(B, H, W, Cin, device="cuda") if has_add else None
    wf, wd = K.planes_split_weight(w, g, 3)
    xs = (B, H, W, Cin)
    res = {}
    if mode == "fwd":
        N = Cout // g
        for t in K._X3_TILES:
            if K._x3_tile_ok(t, N):
                res[t] = timeit(lambda: K.conv2d_fwd_x3(x, wf, bn_stats=True, tile=t, kh=3, pad=1, groups=g))
        t32 = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 1, g, bn_stats=True))
        key = "|".join(["xfwd", "3"] + [str(v) for v in (B, H, W, Cin, Cout, 3, g)])
    else:
        N = Cin // g
        ones, zeros = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        outs = [K.bn_fwd(torch.randn(*xs, device="cuda"), ones, zeros, zeros.clone(), ones.clone(), True, True, want_mask=True)
                for _ in range(nent)]
        xin = [torch.randn(*xs, device="cuda") for _ in range(nent)]
        def fz():
            return K.BnBwdFuse(outs[0][2], [(xin[i], outs[i][1]) for i in range(nent)])
        for t in K._X3_TILES:
            if K._x3_tile_ok(t, N):
                res[t] = timeit(lambda: K.conv2d_dgrad_x3(dy, wd, xs, addend=add, post=fz(), tile=t, kh=kh, pad=pad, groups=g))
        t32 = timeit(lambda: K.conv2d_dgrad(dy, w, xs, add, 1, pad, g, post=fz()))
        key = "|".join(["xdgrad+bn", str(nent), str(int(has_add)), "3"] + [str(v) for v in ((B, H, W, Cin, Cout) if kh == 1 else (B, H, W, Cin, Cout, 3, g))])
    best = min(res, key=res.get)
    ch[key] = best
    tot32 += t32; totx += res[best]
    print("%-52s fp32 %7.1f us | x3 %s -> tile %d %7.1f us (%.2fx)" % (
        ks, t32, " ".join("%d:%6.1f" % kv for kv in sorted(res.items())), best, res[best], t32 / res[best]), flush=True)
print("3x3 layers, sum over the table's shapes: fp32 %.0f us, x3 %.0f us" % (tot32, totx))
doc["choices"] = dict(sorted(ch.items()))
with open(out, "w") as f:
    json.dump(doc, f, indent=0)
print("wrote", out)
