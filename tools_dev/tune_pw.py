"""Re-tunes the fp32 1x1 forward / plain input-gradient entries of the static table for which the persistent pointwise kernel
(tile 4, csrc/conv_pw_persist.h) is a legal candidate: tiles 0-4 timed (forward with the fused BatchNorm statistics, as the
model launches it), 10 launches per candidate, best of two batches.  usage (GPU box): python tools_dev/tune_pw.py [out.json]"""
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


changed = 0
for ks in sorted(k for k in ch if k.startswith(("fwd|0|", "dgrad|0|"))):
    p = ks.split("|")
    mode = p[0]
    B, H, W, Cin, Cout, kh, kw, stride, pad, g = [int(v) for v in p[2:]]
    Kg, Ng = (Cin, Cout) if mode == "fwd" else (Cout, Cin)
    if not K._pw_persist_legal(B * H * W, Kg, Ng, kh, kw, stride, pad, g, True):
        continue
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(1, 1, Cin, Cout, device="cuda") * 0.05
    dy = torch.randn(B, H, W, Cout, device="cuda")
    key = (mode, False, B, H, W, Cin, Cout, kh, kw, stride, pad, g)
    res = {}
    for t in (0, 1, 2, 3, 4):
        if t < 4 and not K._tile_legal(Ng, t):
            continue
        K._tile_cache[key] = t
        if mode == "fwd":
            res[t] = timeit(lambda: K.conv2d_fwd(x, w, None, None, 1, 0, 1, bn_stats=True))
        else:
            res[t] = timeit(lambda: K.conv2d_dgrad(dy, w, (B, H, W, Cin), None, 1, 0, 1))
    K._tile_cache.pop(key, None)
    best = min(res, key=res.get)
    # the persistent kernel only where it wins by more than the timing noise
    if best == 4 and res[4] > 0.97 * min(v for t, v in res.items() if t != 4):
        best = min((t for t in res if t != 4), key=res.get)
    print("%-46s old %d %7.1f us -> %d %7.1f us%s" % (ks, ch[ks], res.get(ch[ks], float("nan")), best, res[best],
                                                      "   (persistent %.1f)" % res[4]), flush=True)
    changed += int(best != ch[ks])
    ch[ks] = best
    del x, dy
    torch.cuda.empty_cache()
# ---- the fused input gradients (BatchNorm-backward epilogue): "dgrad+bn|entries|addend|0|B|H|W|Cin|Cout|1|1|1|0|1"
for ks in sorted(k for k in ch if k.startswith("dgrad+bn|")):
    p = ks.split("|")
    nent, with_add, bf16 = int(p[1]), int(p[2]), int(p[3])
    B, H, W, Cin, Cout, kh, kw, stride, pad, g = [int(v) for v in p[4:]]
    if bf16 or not K._pw_persist_legal(B * H * W, Cout, Cin, kh, kw, stride, pad, g, True):
        continue
    dy = torch.randn(B, H, W, Cout, device="cuda"); w = torch.randn(1, 1, Cin, Cout, device="cuda") * 0.05
    addend = torch.randn(B, H, W, Cin, device="cuda") if with_add else None
    ents = []
    gam, bet = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    x1 = torch.randn(B, H, W, Cin, device="cuda")
    if nent == 2:
        x2 = torch.randn(B, H, W, Cin, device="cuda")
        y2, sv2 = K.bn_fwd(x2, gam, bet, torch.zeros_like(gam), torch.ones_like(gam), True, False)
        _, sv1, mask = K.bn_fwd(x1, gam, bet, torch.zeros_like(gam), torch.ones_like(gam), True, True, residual=y2, want_mask=True)
        ents = [(x1, sv1), (x2, sv2)]
        del y2
    else:
        _, sv1, mask = K.bn_fwd(x1, gam, bet, torch.zeros_like(gam), torch.ones_like(gam), True, True, want_mask=True)
        ents = [(x1, sv1)]
    key = ("dgrad+bn", nent, bool(with_add), False, B, H, W, Cin, Cout, kh, kw, stride, pad, g)
    res = {}
    for t in (0, 1, 2, 3, 4):
        if t < 4 and not K._tile_legal(Cin, t):
            continue
        K._tile_cache[key] = t

        def run():
            post = K.BnBwdFuse(mask, ents)
            return K.conv2d_dgrad(dy, w, (B, H, W, Cin), addend, 1, 0, 1, post=post)
        res[t] = timeit(run)
    K._tile_cache.pop(key, None)
    best = min(res, key=res.get)
    if best == 4 and res[4] > 0.97 * min(v for t, v in res.items() if t != 4):
        best = min((t for t in res if t != 4), key=res.get)
    byts = 4.0 * B * H * W * (Cout + Cin * (2 + nent + with_add))
    print("%-52s old %d %7.1f us -> %d %7.1f us   (persistent %.1f; HBM floor at 6.3 TB/s %.1f us)"
          % (ks, ch[ks], res.get(ch[ks], float("nan")), best, res[best], res[4], byts / 6.3e6), flush=True)
    changed += int(best != ch[ks])
    ch[ks] = best
    del dy, x1, ents, mask, addend
    torch.cuda.empty_cache()
print(changed, "entries changed")
json.dump({"arch": doc.get("arch", "gfx950"), "choices": dict(sorted(ch.items()))}, open(out, "w"), indent=0)
