"""Re-tunes ONLY the plane weight-gradient entries ("pwgrad|...") of the static table against the current candidate plans
(incl. the tap-fused kernel's 64..67) and writes the merged table.  10 launches per candidate, best of two batches, like
tools_dev/tune_table.py.  usage (GPU box): python tools_dev/tune_pwgrad.py [out.json]  (default: in place)"""
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


for ks in sorted(k for k in ch if k.startswith("pwgrad|")):
    np_, B, H, W, Cin, Cout, kh, kw, pad, g = [int(v) for v in ks.split("|")[1:]]
    x = torch.randn(B, H, W, Cin, device="cuda"); dy = torch.randn(B, H, W, Cout, device="cuda")
    xp, dyp = K.planes_split(x, np_), K.planes_split(dy, np_)
    dw = torch.empty(kh, kw, Cin // g, Cout, device="cuda")
    key = ("pwgrad", np_, B, H, W, Cin, Cout, kh, kw, pad, g)
    res = {}
    for plan in K._PWGRAD_PLANS:
        if plan >= 64 and not (kh == 3 and pad == 1 and W <= 63 and 2 * B * H * W * max(Cin, Cout) < (1 << 31)):
            continue
        K._tile_cache[key] = plan
        res[plan] = timeit(lambda: K.conv2d_wgrad_planes(xp, dyp, dw, pad, g))
    K._tile_cache.pop(key, None)
    best = min(res, key=res.get)
    print("%-48s old %3d %7.1f us -> %3d %7.1f us" % (ks, ch[ks], res.get(ch[ks], float("nan")), best, res[best]), flush=True)
    ch[ks] = best
    del x, dy, xp, dyp
    torch.cuda.empty_cache()
json.dump({"arch": doc.get("arch", "gfx950"), "choices": dict(sorted(ch.items()))}, open(out, "w"), indent=0)
