"""Which conv call makes bf16 features deviate?  Runs the backbone forward with bf16 enabled for ONE conv2d_fwd call."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_model_gpu import build
from scouter_amd import kernels as K
m, P, images, labels = build("resnest26d_224")
m.train()
x = images.cuda().float()
orig = K.conv2d_fwd
state = {"i": 0, "target": -1, "shapes": []}
def patched(x_, w, bias=None, addend=None, stride=1, pad=0, groups=1, relu=False, bn_stats=False):
    i = state["i"]; state["i"] += 1
    K.PRECISION = "bf16" if (i == state["target"] or state["target"] == -2) else "fp32"
    if state["target"] == -1: state["shapes"].append((tuple(x_.shape), tuple(w.shape), stride, pad, groups, relu, bn_stats, bias is not None, addend is not None))
    try:
        return orig(x_, w, bias, addend, stride, pad, groups, relu, bn_stats)
    finally:
        K.PRECISION = "fp32"
K.conv2d_fwd = patched
import scouter_amd.nn_hip as nn_hip
nn_hip.K.conv2d_fwd = patched
def feats():
    state["i"] = 0
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    f, _ = m.backbone.features_fwd(x, False, [])
    m.load_state_dict(sd)
    return f.double()
ref = feats()
n = state["i"]
print("conv calls:", n)
for t in list(range(n)) + [-2]:
    state["target"] = t
    f = feats()
    r = float((f - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(t, "%.3e" % r, state["shapes"][t] if t >= 0 else "ALL bf16")
