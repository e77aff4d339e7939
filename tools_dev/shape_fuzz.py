"""Odd batch / image sizes through the whole training step: the default path (plane kernels, fused epilogues, tuned
tiles) against the plain path (SCOUTER_PLANES=0-equivalent, static tiles) -- loss and every gradient must agree to
fp32-rounding level (the forward is bit-identical by construction only for equal tiles; here sign flips may occur, so
the check is loose on single tensors, tight on the loss).  usage: python tools_dev/shape_fuzz.py"""
import argparse, sys
import torch
sys.path.insert(0, '.')
from oracle import torch_oracle as O
from scouter_amd import kernels as K
from scouter_amd.sloter.slot_model import SlotModel

def run(arch, B, H, planes, seed=7):
    C, spc, L = 10, 1, 3
    args = argparse.Namespace(model=arch, pre_trained=False, num_classes=C, dataset="ImageNet", use_slot=True,
                              use_pre=False, grad=False, channel=O.ARCHS[arch]["channel"], slots_per_class=spc,
                              hidden_dim=64, freeze_layers=0, vis=False, vis_id=0, loss_status=1, power=2,
                              to_k_layer=L, lambda_value="1")
    P = O.synth_state(O.state_dict_spec(arch, C, spc, L), seed)
    images, labels = O.synth_batch(B, 3, H, C, seed + 1)
    m = SlotModel(args); m.load_state_dict(P); m.set_planes(planes); m.set_x3(31 if planes else 0); m = m.cuda().train()
    out, losses = m(images.cuda(), labels.cuda())
    losses[0].backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    return float(losses[0]), g

for arch, B, H in [("resnest26d", 1, 224), ("resnest26d", 3, 160), ("resnest26d", 33, 96), ("resnest26d", 5, 256),
                   ("resnest26d", 2, 320), ("resnest50d", 7, 128), ("resnet18", 9, 224), ("resnest26d", 17, 72)]:
    try:
        l1, g1 = run(arch, B, H, 3)
        l0, g0 = run(arch, B, H, 0)
    except Exception as e:
        print(arch, B, H, "FAILED:", repr(e)[:300]); continue
    fin = all(bool(torch.isfinite(v).all()) for v in g1.values())
    rel = sorted(((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-20), k) for k in g0), reverse=True)
    med = rel[len(rel) // 2][0]
    print("%-11s B=%-3d H=%-3d loss %.6f vs %.6f  finite %s  grad rel diff worst %.1e (%s) median %.1e" % (
        arch, B, H, l1, l0, fin, rel[0][0], rel[0][1], med))
