"""The CPU restatement (oracle/torch_oracle.py) against the committed golden vectors that oracle/gen_golden.py
produced from the reference itself.  Runs anywhere (no reference, no GPU needed)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from oracle.gen_golden import (HEAD_CASES, MODEL_CASES, LAMBDA, ENGINE_BATCH, ENGINE_SIZE, head_inputs, model_inputs,
                               grad_digest)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def head_cfg(case):
    C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
    return dict(num_classes=C, slots_per_class=spc, loss_status=ls, power=power, lambda_value=float(LAMBDA))


def oracle_head(case, dtype):
    feat, labels, P = head_inputs(case)
    P = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in P.items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    feat = feat.to(dtype).requires_grad_(True)
    aux = {}
    out, losses = O.head_forward(leaves, feat, labels, head_cfg(case), aux=aux)
    losses[0].backward()
    return out, losses, aux, feat, leaves


@pytest.mark.parametrize("case", list(HEAD_CASES))
def test_head_oracle_matches_reference_fixture(case):
    g = load("head_" + case)
    C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
    out, losses, aux, feat, leaves = oracle_head(case, torch.float32)
    floor = np.abs(g["f32_logits"] - g["f64_logits"]).max()       # the reference's own fp32 noise
    tol = max(2e-6, 3 * floor)
    np.testing.assert_allclose(aux["logits"].detach().numpy(), g["f32_logits"], atol=tol, rtol=1e-5)
    np.testing.assert_allclose(out.detach().numpy(), g["f32_log_probs"], atol=tol, rtol=1e-5)
    np.testing.assert_allclose(aux["attn"].detach().numpy(), g["f32_attn"], atol=max(1e-6, tol), rtol=1e-5)
    np.testing.assert_allclose(float(losses[0]), float(g["f32_loss"]), atol=tol, rtol=1e-5)
    np.testing.assert_allclose(float(losses[1]), float(g["f32_nll"]), atol=tol, rtol=1e-5)
    np.testing.assert_allclose(float(losses[2]), float(g["f32_area"]), atol=1e-6, rtol=1e-5)
    pe = O.posenc_sine(side, side, 64)
    np.testing.assert_array_equal(pe.numpy(), g["f32_pe"])
    np.testing.assert_array_equal(O.vis_maps(aux["attn"], C, spc, 0).shape, g["f32_vis"].shape)
    assert np.abs(O.vis_maps(aux["attn"], C, spc, 0).astype(int) - g["f32_vis"].astype(int)).max() <= 1
    # gradients
    gt = max(1e-6, 30 * floor)
    np.testing.assert_allclose(feat.grad[:, :16].numpy(), g["f32_dfeat_head"], atol=gt, rtol=1e-3)
    np.testing.assert_allclose(leaves["conv1x1.weight"].grad[:, :32].numpy(), g["f32_d_conv_w_head"], atol=gt,
                               rtol=1e-3)
    np.testing.assert_allclose(leaves["conv1x1.bias"].grad.numpy(), g["f32_d_conv_b"], atol=gt, rtol=1e-3)
    for k in leaves:
        if not k.startswith("slot."):
            continue
        ref = g["f32_d_" + k]
        if "to_q" in k:
            assert ref.size == 0 and leaves[k].grad is None     # to_q is unused (slot_attention.py:52-53)
            continue
        np.testing.assert_allclose(leaves[k].grad.numpy(), ref, atol=gt, rtol=2e-3, err_msg=k)


@pytest.mark.parametrize("case", list(HEAD_CASES))
def test_head_oracle_fp64_matches_reference_fp64(case):
    g = load("head_" + case)
    out, losses, aux, _, _ = oracle_head(case, torch.float64)
    spc = HEAD_CASES[case][1]
    # slots_per_class > 1: the reference aggregates into torch.zeros(...) (slot_attention.py:88), which is fp32
    # even in an fp64 run, so its fp64 "truth" logits are fp32-rounded; the restatement sums in native dtype.
    rt = 1e-9 if spc == 1 else 3e-7
    np.testing.assert_allclose(aux["logits"].detach().numpy(), g["f64_logits"], atol=rt, rtol=rt)
    if spc > 1:
        return
    np.testing.assert_allclose(aux["attn"].detach().numpy(), g["f64_attn"], atol=1e-10, rtol=1e-9)
    np.testing.assert_allclose(float(losses[0]), float(g["f64_loss"]), atol=1e-10, rtol=1e-10)


def oracle_model(case, dtype):
    arch, C, spc, L, ls, power, B, H, in_chans, mnist = MODEL_CASES[case]
    spec, P, images, labels = model_inputs(case)
    P = {k: (v.to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
    keys = O.trainable_keys(P)
    leaves = {k: P[k].clone().requires_grad_(True) for k in keys}
    Q = dict(P)
    Q.update(leaves)
    cfg = dict(model=arch, num_classes=C, slots_per_class=spc, loss_status=ls, power=power,
               lambda_value=float(LAMBDA))
    aux = {}
    out, losses = O.slot_model_forward(Q, images.to(dtype), labels, cfg, training=True, aux=aux)
    losses[0].backward()
    with torch.no_grad():
        ev = O.slot_model_forward(Q, images.to(dtype), None, cfg, training=False)
    return out, losses, aux, leaves, Q, ev


@pytest.mark.parametrize("case", ["resnet18_mnist_64", "resnest26d_96", "resnest50d_64_spc3"])
def test_model_oracle_matches_reference_fixture(case):
    g = load("model_" + case)
    out, losses, aux, leaves, Q, ev = oracle_model(case, torch.float32)
    floor = np.abs(g["f32_log_probs"] - g["f64_log_probs"]).max()
    tol = max(5e-6, 3 * floor)
    np.testing.assert_allclose(out.detach().numpy(), g["f32_log_probs"], atol=tol, rtol=1e-4)
    np.testing.assert_allclose(aux["attn"].detach().numpy(), g["f32_attn"], atol=tol, rtol=1e-4)
    np.testing.assert_allclose(float(losses[0]), float(g["f32_loss"]), atol=tol, rtol=1e-4)
    np.testing.assert_allclose(ev.numpy(), g["f32_eval_log_probs"], atol=max(tol, 1e-4), rtol=1e-3)
    keys = [str(k) for k in g["f32_grad_keys"]]
    assert sorted(keys) == sorted(leaves.keys())
    assert sorted(str(k) for k in g["f32_unused"]) == ["slot.to_q.0.bias", "slot.to_q.0.weight"]
    dig = {k: d for k, d in zip(keys, g["f64_grad_digest"])}
    # gradients are compared against the reference's fp64 run; fp32 autograd noise through 18-50 layers ~1e-3 rel
    for k in keys:
        mine = grad_digest(leaves[k].grad)
        if k.endswith("conv2.fc1.bias"):    # bias in front of a train-mode BN: exact gradient is 0
            assert abs(mine[2:]).max() < 1e-3 and abs(dig[k][2:]).max() < 1e-9
            continue
        scale = max(dig[k][1] / max(leaves[k].numel(), 1), 1e-8)
        # absolute floors: e.g. conv2.fc1.bias feeds a train-mode BN, its true gradient is 0 (fp32 gives ~1e-5)
        assert abs(mine[0] - dig[k][0]) <= 2e-2 * dig[k][1] + 2e-4, k
        assert abs(mine[1] - dig[k][1]) <= 2e-2 * dig[k][1] + 2e-4, k
        np.testing.assert_allclose(mine[2:], dig[k][2:], atol=50 * scale * 2e-2 + 2e-5, rtol=5e-2, err_msg=k)
    bn_keys = [k for k in Q if k.endswith("running_mean") or k.endswith("running_var")]
    for k, d in zip(bn_keys, g["f32_bn_digest"]):
        np.testing.assert_allclose(grad_digest(Q[k]), d, rtol=1e-4, atol=1e-5, err_msg=k)


def test_model_oracle_fp64_matches_reference_fp64_full_size():
    """resnest26d at the BASELINE input size 224x224 (batch 6): fp64 restatement == fp64 reference."""
    case = "resnest26d_224"
    g = load("model_" + case)
    out, losses, aux, leaves, Q, ev = oracle_model(case, torch.float64)
    np.testing.assert_allclose(out.detach().numpy(), g["f64_log_probs"], atol=1e-8, rtol=1e-8)
    np.testing.assert_allclose(aux["attn"].detach().numpy(), g["f64_attn"], atol=1e-9, rtol=1e-8)
    keys = [str(k) for k in g["f32_grad_keys"]]
    for k, d in zip(keys, g["f64_grad_digest"]):
        np.testing.assert_allclose(grad_digest(leaves[k].grad), d, rtol=1e-6, atol=1e-10, err_msg=k)


def test_engine_oracle_matches_reference_fixture():
    """2 training steps + eval of config 1 (batch 16, 128x128): MetricLog record entries and post-step parameters."""
    g = load("engine_mnist")
    spec = O.state_dict_spec("resnet18", 10, 1, 1, in_chans=1, mnist_stem=True)
    P = O.synth_state(spec, 300)
    cfg = dict(model="resnet18", num_classes=10, slots_per_class=1, loss_status=1, power=1, lambda_value=1.0)
    tr = O.OracleTrainer(P, cfg, lr=1e-4)
    batches = [O.synth_batch(ENGINE_BATCH, 1, ENGINE_SIZE, 10, 310 + i) for i in range(2)]
    run = np.zeros(4)
    for img, lab in batches:
        out, losses, acc, _ = tr.step(img, lab)
        run += np.array([losses[0], acc, losses[1], losses[2]])
    rec = np.round(run / 2, 3)
    np.testing.assert_allclose(rec, g["record_train"], atol=1.01e-3)
    run = np.zeros(4)
    with torch.no_grad():
        for img, lab in batches:
            out, losses = O.slot_model_forward(tr.P, img, lab, cfg, training=False)
            acc = (out.argmax(1) == lab).sum().float().item() / lab.size(0)
            run += np.array([float(losses[0]), acc, float(losses[1]), float(losses[2])])
    np.testing.assert_allclose(np.round(run / 2, 3), g["record_val"], atol=1.01e-3)
    for k, d in zip(g["param_keys"], g["param_digest"]):
        np.testing.assert_allclose(grad_digest(tr.P[str(k)]), d, rtol=2e-4, atol=2e-5, err_msg=str(k))


def test_fc_baseline_oracle_matches_reference_fixture():
    """use_slot=False (slot_model.py:75-77,123-125): backbone + global average pool + fc."""
    g = load("model_fc_resnet18_mnist_64")
    spec = O.state_dict_spec("resnet18", 10, 1, 1, in_chans=1, mnist_stem=True, use_slot=False)
    P = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in O.synth_state(spec, 400).items()}
    keys = [k for k in O.trainable_keys(P)]
    leaves = {k: P[k].clone().requires_grad_(True) for k in keys}
    Q = dict(P); Q.update(leaves)
    images, labels = O.synth_batch(4, 1, 64, 10, 401)
    out, losses = O.fc_model_forward(Q, images.double(), labels, "resnet18", training=True)
    losses[0].backward()
    np.testing.assert_allclose(out.detach().numpy(), g["f64_log_probs"], atol=1e-9)
    for k, d in zip(g["f64_grad_keys"], g["f64_grad_digest"]):
        np.testing.assert_allclose(grad_digest(leaves[str(k)].grad), d, rtol=1e-6, atol=1e-10, err_msg=str(k))
