"""The hand-derived xSlot backward (oracle/xslot_manual.py -- the maths of the HIP backward kernel) against
torch autograd of the restatement (oracle/torch_oracle.py), in fp64."""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from oracle import xslot_manual as M


@pytest.mark.parametrize("C,spc,N,L,ls,power", [(10, 1, 49, 3, 1, 2), (4, 3, 9, 1, -1, 1), (7, 2, 81, 2, 1, 2)])
def test_manual_backward_matches_autograd(C, spc, N, L, ls, power):
    rng = np.random.default_rng(7)
    B, d, S = 3, 64, C * spc
    lam = 0.7
    spec = {k: v for k, v in O.state_dict_spec("resnet18", C, spc, L).items() if k.startswith("slot.")}
    P = {k: v.double() for k, v in O.synth_state(spec, 11).items()}
    X = torch.from_numpy(np.maximum(rng.standard_normal((B, N, d)), 0.0))
    side = int(N ** 0.5)
    PE = O.posenc_sine(side, side, d, torch.float64).reshape(d, N).t().contiguous()
    y = torch.from_numpy(rng.integers(0, C, B))

    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if "to_q" not in k}
    Xl = X.clone().requires_grad_(True)
    aux = {}
    logits, term = O.xslot_forward(leaves | {k: v for k, v in P.items() if "to_q" in k}, Xl + PE, Xl, C, spc, ls,
                                   power, aux=aux)
    logp = torch.log_softmax(logits, 1)
    loss = torch.nn.functional.nll_loss(logp, y) + lam * term
    loss.backward()

    W = dict(to_k_w=np.stack([P[f"slot.to_k.{2*l}.weight"].numpy() for l in range(L)]),
             to_k_b=np.stack([P[f"slot.to_k.{2*l}.bias"].numpy() for l in range(L)]),
             slots0=P["slot.initial_slots"][0].numpy(), w_ih=P["slot.gru.weight_ih_l0"].numpy(),
             w_hh=P["slot.gru.weight_hh_l0"].numpy(), b_ih=P["slot.gru.bias_ih_l0"].numpy(),
             b_hh=P["slot.gru.bias_hh_l0"].numpy())
    lg, area_sum, _ = M.forward(W, X.numpy(), PE.numpy(), C, spc, ls)
    np.testing.assert_allclose(lg, logits.detach().numpy(), rtol=1e-10, atol=1e-10)
    area = area_sum / (B * S * N)
    # dL/dlogits and dL/d(area_sum) exactly as the loss kernel computes them
    sm = torch.softmax(logits.detach(), 1).numpy()
    onehot = np.eye(C)[y.numpy()]
    dlogits = (sm - onehot) / B
    g_area = lam * power * area ** (power - 1) / (B * S * N)
    g = M.backward(W, X.numpy(), PE.numpy(), C, spc, dlogits, g_area, ls)

    def chk(a, b, name):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-7, atol=1e-10, err_msg=name)
    chk(g["dX"], Xl.grad, "dX")
    chk(g["slots0"], leaves["slot.initial_slots"].grad[0], "slots0")
    chk(g["w_ih"], leaves["slot.gru.weight_ih_l0"].grad, "w_ih")
    chk(g["w_hh"], leaves["slot.gru.weight_hh_l0"].grad, "w_hh")
    chk(g["b_ih"], leaves["slot.gru.bias_ih_l0"].grad, "b_ih")
    chk(g["b_hh"], leaves["slot.gru.bias_hh_l0"].grad, "b_hh")
    for l in range(L):
        chk(g["to_k_w"][l], leaves[f"slot.to_k.{2*l}.weight"].grad, f"to_k_w{l}")
        chk(g["to_k_b"][l], leaves[f"slot.to_k.{2*l}.bias"].grad, f"to_k_b{l}")
