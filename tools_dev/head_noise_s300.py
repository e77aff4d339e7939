"""resnest50d + 100 x 3 slots (S = 300, the ill-conditioned head of SURVEY fact 10), batch 2, 224 x 224: |HIP - fp64| of the
log-probabilities next to |PyTorch fp32 - fp64| over several seeds, for the forward-option sets given as X3:HALO pairs.
    python tools_dev/head_noise_s300.py 15:2 63:3"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_oracle as O
from scouter_amd import kernels as K
from scouter_amd.sloter.slot_model import SlotModel

arch, C, spc, B, H, L = "resnest50d", 100, 3, 2, 224, 3
sets = [tuple(int(v) for v in s.split(":")) for s in sys.argv[1:]] or [(15, 2), (63, 3)]
torch.set_num_threads(min(32, os.cpu_count() or 1))
for seed in (900, 1900, 2900, 3900, 4900, 5900):
    args = argparse.Namespace(model=arch, pre_trained=False, num_classes=C, dataset="ImageNet", use_slot=True, use_pre=False,
                              grad=False, channel=2048, slots_per_class=spc, hidden_dim=64, freeze_layers=0, vis=False, vis_id=0,
                              loss_status=1, power=2, to_k_layer=L, lambda_value="1")
    P = O.synth_state(O.state_dict_spec(arch, C, spc, L), seed)
    images, labels = O.synth_batch(B, 3, H, C, seed + 1)
    cfg = dict(model=arch, num_classes=C, slots_per_class=spc, loss_status=1, power=2, lambda_value=1.0)
    with torch.no_grad():
        aux = {}
        ref = O.slot_model_forward({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in P.items()},
                                   images.double(), labels, cfg, training=True, aux=aux)[0]
        floors = []
        for t in (8, 16, 32):
            torch.set_num_threads(t)
            r32 = O.slot_model_forward({k: v.clone() for k, v in P.items()}, images, labels, cfg, training=True)[0]
            floors.append(float((r32.double() - ref).abs().max()))
    row = []
    for x3, halo in sets:
        K.HALO_TILE = halo
        m = SlotModel(args); m.load_state_dict(P); m = m.cuda().train(); m.set_x3(x3)
        with torch.no_grad():
            out, _ = m(images.cuda(), labels.cuda())
        err = float((out.cpu().double() - ref).abs().max())
        erra = float((m.slot.last_attn.cpu().double() - aux["attn"]).abs().max())
        row.append("x3=%d halo=%d: err %.3g attn %.3g" % (x3, halo, err, erra))
        del m
    print("seed %d: torch fp32 floors (8/16/32 threads) %s | %s" % (seed, ["%.3g" % f for f in floors], " | ".join(row)), flush=True)
