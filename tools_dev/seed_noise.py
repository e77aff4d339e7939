"""Rounding noise of the HIP path as a DISTRIBUTION over the five seeds of tests/golden/model_resnest26d_224_seeds.npz, forward
(log-probabilities) and head gradients, for the default path and every forward-changing option -- the data the bounds of
tests/test_model_gpu.py::test_rounding_noise_over_five_seeds / ::test_gradient_noise_over_five_seeds were calibrated on.

    python tools_dev/seed_noise.py            (on an MI355X)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_oracle as O                                              # noqa: E402
from oracle.gen_golden import MODEL_CASES, LAMBDA, SEED_CASE, model_inputs       # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def run(switches, g):
    from scouter_amd.sloter.slot_model import SlotModel
    from scouter_amd import kernels as K
    arch, C, spc, L, ls, power, B, H, in_chans, mnist = MODEL_CASES[SEED_CASE]
    K.HALO_TILE = switches.get("halo", 2)
    keys = [str(k) for k in g["grad_keys"]]
    head = [str(k) for k in g["head_keys"]]
    sizes = [int(s) for s in g["head_sizes"]]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    fwd, grad_worst, grad_med, per_tensor = [], [], [], []
    for i, seed in enumerate(g["seeds"]):
        args = argparse.Namespace(model=arch, pre_trained=False, num_classes=C, dataset="ImageNet", use_slot=True, use_pre=False,
                                  grad=False, channel=O.ARCHS[arch]["channel"], slots_per_class=spc, hidden_dim=64,
                                  freeze_layers=0, vis=False, vis_id=0, loss_status=ls, power=power, to_k_layer=L,
                                  lambda_value=LAMBDA)
        spec, P, images, labels = model_inputs(SEED_CASE, int(seed))
        m = SlotModel(args)
        m.load_state_dict(P)
        m = m.cuda().train()
        if "x3" in switches:
            m.set_x3(switches["x3"])
        out, losses = m(images.cuda(), labels.cuda())
        losses[0].backward()
        torch.cuda.synchronize()
        err = float(np.abs(out.detach().cpu().numpy().astype(np.float64) - g["f64_log_probs"][i]).max())
        floors = np.abs(g["f32_log_probs"][i].astype(np.float64) - g["f64_log_probs"][i][None]).max((1, 2))
        fwd.append(err / float(np.exp(np.log(floors).mean())))
        named = dict(m.named_parameters())
        rs = []
        for j, k in enumerate(head):
            ref = g["f64_head_grads"][i][offs[j]:offs[j + 1]].astype(np.float64)
            mine = named[k].grad.detach().cpu().double().flatten().numpy()
            e = float(np.abs(mine - ref).max())
            e32 = g["f32_grad_maxdev"][i][:, keys.index(k)]
            scale = float(g["f64_grad_absmax"][i][keys.index(k)])
            rs.append(e / max(float(np.exp(np.log(e32).mean())), 5e-4 * scale))
        per_tensor.append(rs)
        grad_worst.append(max(rs))
        grad_med.append(float(np.median(rs)))
        del m
    gm = lambda v: float(np.exp(np.log(v).mean()))
    print("%-22s fwd ratios %s gm %.2f | head-gradient worst tensor per seed %s gm %.2f max %.2f | median tensor %s gm %.2f"
          % (switches or "(default)", ["%.2f" % r for r in fwd], gm(fwd), ["%.2f" % r for r in grad_worst], gm(grad_worst),
             max(grad_worst), ["%.2f" % r for r in grad_med], gm(grad_med)), flush=True)
    pt = np.array(per_tensor)                         # [seed][tensor]
    print("    per tensor gm over seeds:", " ".join("%s=%.2f" % (k.replace("slot.", ""), gm(pt[:, j])) for j, k in enumerate(head)),
          flush=True)


if __name__ == "__main__":
    import __graft_entry__ as G
    G.build()
    g = np.load(os.path.join(GOLD, "model_%s_seeds.npz" % SEED_CASE))
    sets = [{}, {"x3": 31}, {"x3": 47}, {"x3": 63}, {"halo": 3}, {"x3": 63, "halo": 3}]
    if len(sys.argv) > 1:            # e.g. 127:3 79:2  (x3:halo pairs)
        sets = [dict(x3=int(a.split(":")[0]), halo=int(a.split(":")[1])) for a in sys.argv[1:]]
    for sw in sets:
        run(sw, g)
