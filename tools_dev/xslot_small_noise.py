"""Forward error of the small-S xSlot kernel and of the 32-slot-tile kernel against the fp64 oracle, next to what plain fp32
PyTorch loses on the same inputs, over several seeds.  usage: python tools_dev/xslot_small_noise.py [B S N T L spc] [seeds]"""
import os, sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_xslot_small_gpu as Tt
from oracle import torch_oracle as O
a = [int(v) for v in sys.argv[1:]]
B, S, N, T, L, spc = (a + [5, 16, 64, 3, 2, 1][len(a):])[:6]
nseed = a[6] if len(a) > 6 else 8
for seed in range(nseed):
    P, X, PE, wl, ga = Tt._inputs(B, S, N, T, L, spc, seed=seed)
    def orc(dt):
        Q = {k: v.to(dt) for k, v in P.items()}
        aux = {}
        lg, _ = O.xslot_forward(Q, (X + PE).to(dt), X.to(dt), S // spc, spc, 1, 1, iters=T, aux=aux)
        return lg, aux["attn"]
    l64, a64 = orc(torch.float64); l32, a32 = orc(torch.float32)
    fs, _ = Tt._run(P, X, PE, wl, ga, S, T, L, spc, True)
    ft, _ = Tt._run(P, X, PE, wl, ga, S, T, L, spc, False)
    e = lambda t, r: float((t.cpu().double() - r).abs().max())
    print("seed %d  logits: small %.2e tiles %.2e torch-fp32 %.2e | attn: small %.2e tiles %.2e torch-fp32 %.2e | max|logit| %.2f" % (
        seed, e(fs["logits"], l64), e(ft["logits"], l64), e(l32, l64), e(fs["attn"], a64), e(ft["attn"], a64), e(a32, a64),
        float(l64.abs().max())))
