"""Dev check: deviation of the HIP path from the fp64 oracle next to plain fp32 PyTorch's deviation on the same
inputs, over a few seeds / shapes.   usage: python tools_dev/fp32_floor_check.py"""
import argparse, sys, torch
sys.path.insert(0, '.')
from oracle import torch_oracle as O
from scouter_amd.sloter.slot_model import SlotModel
for arch, H, C, spc in (("resnest26d", 96, 10, 1), ("resnest26d", 128, 10, 1), ("resnet18", 96, 10, 1), ("resnest26d", 96, 20, 3)):
    for seed in (200, 300, 400):
        L, B = 3, 4
        args = argparse.Namespace(model=arch, pre_trained=False, num_classes=C, dataset="ImageNet", use_slot=True,
                                  use_pre=False, grad=False, channel=2048 if arch != "resnet18" else 512, slots_per_class=spc,
                                  hidden_dim=64, freeze_layers=0, vis=False, vis_id=0, loss_status=1, power=2, to_k_layer=L,
                                  lambda_value="1")
        P = O.synth_state(O.state_dict_spec(arch, C, spc, L), seed)
        images, labels = O.synth_batch(B, 3, H, C, seed + 1)
        m = SlotModel(args); m.load_state_dict(P); m = m.cuda().train()
        out, losses = m(images.cuda(), labels.cuda())
        cfg = dict(model=arch, num_classes=C, slots_per_class=spc, loss_status=1, power=2, lambda_value=1.0)
        Pd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
        ref, _ = O.slot_model_forward(Pd, images.double(), labels, cfg, training=True)
        ref32, _ = O.slot_model_forward({k: v.clone() for k, v in P.items()}, images.float(), labels, cfg, training=True)
        e = float((out.detach().cpu().double() - ref).abs().max()); f = float((ref32.double() - ref).abs().max())
        print("%-11s %3dpx C=%d spc=%d seed %d: hip %.2e  torch-fp32 %.2e  ratio %.2f" % (arch, H, C, spc, seed, e, f, e / f))
