"""HIP bf16 backbone features vs the bf16-emulating oracle, with rounding enabled for the first K conv calls only."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_model_gpu import build, MODEL_CASES, model_inputs
from oracle import torch_oracle as O
from scouter_amd import kernels as K
case = "resnest26d_224"
m, P, images, labels = build(case)
m.train()
x = images.cuda().float()
orig = K.conv2d_fwd
st = {"i": 0, "k": 0}
def patched(x_, w, bias=None, addend=None, stride=1, pad=0, groups=1, relu=False, bn_stats=False):
    i = st["i"]; st["i"] += 1
    K.PRECISION = "bf16" if i < st["k"] else "fp32"
    try:
        return orig(x_, w, bias, addend, stride, pad, groups, relu, bn_stats)
    finally:
        K.PRECISION = "fp32"
import scouter_amd.nn_hip as nn_hip
nn_hip.K.conv2d_fwd = patched
oconv = O._conv
ost = {"i": 0}
def oc(x_, w, bias=None, stride=1, pad=0, dilation=1, groups=1):
    i = ost["i"]; ost["i"] += 1
    O.CONV_INPUT_ROUNDING = "bf16" if i < st["k"] else None
    try:
        return oconv(x_, w, bias, stride, pad, dilation, groups)
    finally:
        O.CONV_INPUT_ROUNDING = None
O._conv = oc
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
spec, P0, _, _ = model_inputs(case)
for k in [0, 1, 2, 3, 4, 5, 8, 12, 20, 47]:
    st["k"] = k; st["i"] = 0; ost["i"] = 0
    m.load_state_dict(sd0)
    f, _ = m.backbone.features_fwd(x, False, [])
    Pd = {kk: (v.double() if v.dtype.is_floating_point else v.clone()) for kk, v in P0.items()}
    fo = O.backbone_features(Pd, images.double(), "resnest26d", True)
    fh = f.permute(0, 3, 1, 2).double().cpu()
    print("first %2d convs rounded: rel rms HIP vs oracle %.3e   (HIP calls %d, oracle calls %d)" % (
        k, float((fh - fo).pow(2).mean().sqrt() / fo.pow(2).mean().sqrt()), st["i"], ost["i"]))
