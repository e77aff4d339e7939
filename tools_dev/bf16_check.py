"""bf16 mode vs fp32 mode on the full-size fixture: feature / log-prob / gradient deviations."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_model_gpu import build, GOLD
case = "resnest26d_224"
g = np.load(os.path.join(GOLD, "model_%s.npz" % case))
res = {}
for prec in ("fp32", "bf16"):
    m, P, images, labels = build(case)
    m.set_precision(prec)
    m.train()
    feat, _ = m.backbone.features_fwd(images.cuda().float(), False, [])
    out, (loss, nll, area) = m(images.cuda(), labels.cuda())
    loss.backward()
    res[prec] = (feat.double().cpu(), out.detach().double().cpu(), float(loss), {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None})
f32, fb = res["fp32"][0], res["bf16"][0]
print("features: scale %.3g  max|bf16-fp32| %.3g  rel rms %.3g" % (f32.abs().max(), (fb - f32).abs().max(), ((fb - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt())))
print("log_probs fp64 ref range [%.3g, %.3g]; |fp32-ref| %.3g  |bf16-ref| %.3g  |bf16-fp32| %.3g" % (g["f64_log_probs"].min(), g["f64_log_probs"].max(), np.abs(res["fp32"][1].numpy() - g["f64_log_probs"]).max(), np.abs(res["bf16"][1].numpy() - g["f64_log_probs"]).max(), (res["bf16"][1] - res["fp32"][1]).abs().max()))
print("loss fp32 %.5f bf16 %.5f ref %.5f" % (res["fp32"][2], res["bf16"][2], float(g["f64_loss"])))
cos = []
for k, a in res["bf16"][3].items():
    b = res["fp32"][3][k]
    if a.numel() >= 4096:
        cos.append((float((a.flatten() @ b.flatten()) / (a.norm() * b.norm() + 1e-300)), k))
cos.sort()
print("gradient cosine bf16 vs fp32: min %.4f (%s)  median %.5f" % (cos[0][0], cos[0][1], np.median([c for c, _ in cos])))
print(cos[:5])
