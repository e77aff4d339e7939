"""Dev check: the xSlot backward must not depend on what the shared workspace / output buffers held before the launch.
Runs it on poisoned (NaN / huge) and zeroed memory and compares the results bit for bit.
usage: python tools_dev/xslot_bwd_determinism.py [B S N T]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scouter_amd import kernels as K

B, S, N, T = (int(v) for v in (sys.argv[1:5] + [4, 300, 49, 3][len(sys.argv) - 1:]))
spc, d, L = 3, 64, 3
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
X, PE = r(B, N, d).relu_(), r(N, d) * 0.3
tw, tb = [r(d, d) * 0.1 for _ in range(L)], [r(d) * 0.1 for _ in range(L)]
s0 = r(S, d).abs() * 0.5
wih, whh, bih, bhh = r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1
fwd = K.xslot_fwd(X, PE, tw, tb, s0, wih, whh, bih, bhh, spc, T, 1)
dl = r(B, S // spc)
ga = torch.ones(1, device=dev) * 0.01
runs = []
keep = []
for fill in (0.0, float("nan"), 1e30, 0.0, float("nan")) * 4:
    ws = K.workspace(1, dev)
    ws.view(torch.float32).fill_(fill)
    junk = torch.empty(64 << 20, device=dev).fill_(fill)      # whatever torch.empty hands out next
    del junk
    keep.append(torch.empty(int(torch.randint(1, 1 << 20, (1,))), device=dev))       # shift what torch.empty hands out next
    out = K.xslot_bwd(X, PE, tw, s0, wih, whh, bih, bhh, fwd, dl, ga, spc, T, 1)
    torch.cuda.synchronize()
    runs.append({k: v.clone() for k, v in out.items()})
ok = True
for k in runs[0]:
    for i, o in enumerate(runs[1:], 1):
        same = torch.equal(torch.nan_to_num(runs[0][k], nan=123.0), torch.nan_to_num(o[k], nan=123.0))
        if not same:
            ok = False
            diff = (runs[0][k] - o[k]).abs()
            print("MISMATCH", k, "run", i, "max", float(diff.nan_to_num(nan=1e9).max()), "nan:", int(o[k].isnan().sum()))
print("deterministic" if ok else "NOT deterministic", "B S N T =", B, S, N, T)
