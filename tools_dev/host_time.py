"""Pure host cost of one training step (zero_grad + forward + loss + backward + AdamW): tiny images so that the GPU is never the
limiter -- how far the eager launch chain is from being host-bound at the real step times (config 2: 15.5 ms, config 5: 37 ms).
usage: python tools_dev/host_time.py"""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_model_gpu as T
from scouter_amd.optim import FusedAdamW
for name, arch, C, spc, prec in [("config 2 (resnest26d, 10 slots, fp32)", "resnest26d", 10, 1, "fp32"), ("config 5 (resnest50d, 300 slots, bf16)", "resnest50d", 100, 3, "bf16")]:
    m, P, images, labels, cfg = T._synthetic_model(arch, C, spc, 3, 4, 64, 77)
    if prec == "bf16":
        m.set_precision("bf16")
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    x, y = images.cuda(), labels.cuda()
    def step():
        opt.zero_grad(); out, losses = m(x, y); losses[0].backward(m.loss_seed(losses[0])); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, (t1 - t0) / 20 * 1e3)
    print("%-44s host enqueue time per step (4 x 64 x 64 images): %.2f ms" % (name, best))
