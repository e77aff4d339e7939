"""Spread of tests/test_engine_gpu.py::test_training_learns_a_separable_task over model seeds, with the small-S xSlot kernels
on and off (SCOUTER_XSLOT_SMALL): is a low final accuracy a property of the trajectory (chaotic first hundred steps) or of a kernel?
usage: python tools_dev/separable_task_spread.py [nseeds]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from scouter_amd.optim import FusedAdamW
from scouter_amd.sloter.slot_model import SlotModel
from scouter_amd.train import get_args_parser

def run(seed, small, precision="fp32", steps=160):
    os.environ["SCOUTER_XSLOT_SMALL"] = "1" if small else "0"
    args = get_args_parser().parse_args(["--dataset", "MNIST", "--model", "resnet18", "--channel", "512", "--img_size", "64",
                                         "--num_classes", "4", "--slots_per_class", "1", "--pre_trained", "false",
                                         "--lambda_value", "0.1", "--precision", precision])
    for name, typ in (("num_classes", int), ("lambda_value", float), ("power", int), ("slots_per_class", int)):
        setattr(args, name, typ(getattr(args, name)))
    torch.manual_seed(seed)
    model = SlotModel(args).cuda().train()
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    g = torch.Generator().manual_seed(7)
    def batch(n=32):
        y = torch.randint(0, 4, (n,), generator=g)
        x = torch.randn(n, 1, 64, 64, generator=g) * 0.5
        for i, c in enumerate(y.tolist()):
            r, q = divmod(c, 2)
            x[i, 0, 32 * r:32 * r + 32, 32 * q:32 * q + 32] += 1.5
        return x.cuda(), y.cuda()
    first, last, acc = [], [], []
    for it in range(steps):
        x, y = batch()
        opt.zero_grad()
        out, losses = model(x, y)
        losses[0].backward()
        opt.step()
        (first if it < 5 else last).append(float(losses[1].detach()))
        if it >= steps - 20:
            acc.append(float((out.argmax(1) == y).float().mean()))
    return np.mean(first), np.mean(last[-10:]), np.mean(acc)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for seed in range(3, 3 + n):
    for small in (False, True):
        f, l, a = run(seed, small)
        print("seed %d small=%d: NLL %.3f -> %.3f, accuracy %.3f" % (seed, small, f, l, a), flush=True)
if len(sys.argv) > 2:
    for seed in (3, 4, 8):
        f, l, a = run(seed, True, "bf16")
        print("bf16 seed %d small=1: NLL %.3f -> %.3f, accuracy %.3f" % (seed, f, l, a), flush=True)
