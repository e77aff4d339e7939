"""TEST INFRASTRUCTURE ONLY -- explicit NumPy forward + hand-derived backward of the xSlot head.

This is the maths the fused HIP kernels implement (scouter_amd/csrc/xslot_fwd.hip, xslot_bwd.hip), written
out operation by operation so that the derivation can be checked against torch autograd of the restatement in
``oracle/torch_oracle.py`` (tests/test_oracle_manual_backward.py) before it is trusted on the GPU.

Forward (reference sloter/utils/slot_attention.py:44-96, per image; X = relu(conv1x1(F)), P = sine PE):
    K   = to_k(X + P)
    s_0 = initial_slots
    for t = 1..T:   D = s_{t-1} K^T * d^-1/2 ; r_i = sum_j D_ij ; tau = sum_i r_i
                    A = sigmoid(D / r_i * tau) ; U = A X / d ; s_t = GRU(U, s_{t-1})   (s_T unused)
    logits_c = ls * sum_{s in c} sum_k U_T[s,k] ; area = mean(A_T) ; term = area ** power
Backward of the normaliser:  G = dL/dZ = dL/dA * A(1-A),  g_i = sum_j G_ij D_ij,
    dL/dD_ij = G_ij tau/r_i - g_i tau/r_i^2 + sum_i' g_i'/r_i'.
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def forward(W, X, PE, num_classes, spc, loss_status=1, T=3):
    """W: dict of numpy arrays (to_k_w[L,d,d], to_k_b[L,d], slots0[S,d], w_ih[3d,d], w_hh[3d,d], b_ih, b_hh).
    X: [B,N,d] post-ReLU tokens, PE: [N,d].  Returns logits[B,C], area_sum (sum of A_T), saved dict."""
    B, N, d = X.shape
    L = W["to_k_w"].shape[0]
    H = [X + PE[None]]                       # H[l] = input of layer l
    for l in range(L):
        z = H[-1] @ W["to_k_w"][l].T + W["to_k_b"][l]
        H.append(np.maximum(z, 0.0) if l < L - 1 else z)
    K = H[-1]
    S = W["slots0"].shape[0]
    s = np.broadcast_to(W["slots0"], (B, S, d)).copy()
    scale = d ** -0.5
    states, U = [], None
    for t in range(T):
        states.append(s)
        D = np.einsum("bid,bjd->bij", s, K) * scale
        r = D.sum(2, keepdims=True)
        tau = r.sum(1, keepdims=True)
        A = _sigmoid(D / r * tau)
        U = np.einsum("bij,bjd->bid", A, X) / d
        if t < T - 1:
            s = gru_forward(W, U, s)[0]
    Uc = U.reshape(B, num_classes, spc, d).sum(2)
    logits = loss_status * Uc.sum(2)
    saved = dict(H=H, K=K, states=states, A_last=A)
    return logits, A.sum(), saved


def gru_forward(W, u, h):
    d = h.shape[-1]
    gi = u @ W["w_ih"].T + W["b_ih"]
    gh = h @ W["w_hh"].T + W["b_hh"]
    r = _sigmoid(gi[..., :d] + gh[..., :d])
    z = _sigmoid(gi[..., d:2 * d] + gh[..., d:2 * d])
    hn = gh[..., 2 * d:]
    n = np.tanh(gi[..., 2 * d:] + r * hn)
    return (1 - z) * n + z * h, (r, z, n, hn)


def backward(W, X, PE, num_classes, spc, dlogits, g_area, loss_status=1, T=3):
    """dlogits: [B,C] = dL/dlogits;  g_area: scalar = dL/d(sum of A_T) (uniform over every element of A_T).
    Returns grads dict: dX[B,N,d] (wrt the post-ReLU tokens), to_k_w, to_k_b, slots0, w_ih, w_hh, b_ih, b_hh."""
    B, N, d = X.shape
    L = W["to_k_w"].shape[0]
    S = W["slots0"].shape[0]
    scale = d ** -0.5
    _, _, saved = forward(W, X, PE, num_classes, spc, loss_status, T)
    K, states, H = saved["K"], saved["states"], saved["H"]
    dX = np.zeros_like(X)
    dK = np.zeros_like(K)
    g = dict(w_ih=np.zeros_like(W["w_ih"]), w_hh=np.zeros_like(W["w_hh"]),
             b_ih=np.zeros_like(W["b_ih"]), b_hh=np.zeros_like(W["b_hh"]))
    # logits_c = ls * sum_{s in c} sum_k U_T[s,k]  ->  dU_T[s,k] = ls * dlogits[c(s)]
    dU = np.repeat(loss_status * dlogits, spc, axis=1)[:, :, None] * np.ones((1, 1, d))
    ds = np.zeros((B, S, d))                      # dL/ds_t flowing into iteration t+1's inputs
    for t in range(T - 1, -1, -1):
        s_prev = states[t]
        # ---- recompute the iteration's forward from s_{t-1}
        D = np.einsum("bid,bjd->bij", s_prev, K) * scale
        r = D.sum(2, keepdims=True)
        tau = r.sum(1, keepdims=True)
        A = _sigmoid(D / r * tau)
        U = np.einsum("bij,bjd->bid", A, X) / d
        dh_prev = np.zeros_like(s_prev)
        if t < T - 1:
            # ---- GRU backward: s_t = GRU(U, s_prev), incoming ds (= dL/ds_t)
            _, (rg, zg, ng, hn) = gru_forward(W, U, s_prev)
            dn = ds * (1 - zg)
            dz = ds * (s_prev - ng)
            dh_prev = ds * zg
            da_n = dn * (1 - ng * ng)
            dr = da_n * hn
            da_r = dr * rg * (1 - rg)
            da_z = dz * zg * (1 - zg)
            dgi = np.concatenate([da_r, da_z, da_n], axis=-1)
            dgh = np.concatenate([da_r, da_z, da_n * rg], axis=-1)
            dU = dgi @ W["w_ih"]
            dh_prev = dh_prev + dgh @ W["w_hh"]
            g["w_ih"] += np.einsum("bsg,bsd->gd", dgi, U)
            g["w_hh"] += np.einsum("bsg,bsd->gd", dgh, s_prev)
            g["b_ih"] += dgi.sum((0, 1))
            g["b_hh"] += dgh.sum((0, 1))
        # ---- U = A X / d
        dA = np.einsum("bid,bjd->bij", dU, X) / d
        if t == T - 1:
            dA = dA + g_area
        dX += np.einsum("bij,bid->bjd", A, dU) / d
        # ---- A = sigmoid(Z), Z = D / r * tau
        G = dA * A * (1 - A)
        gi_ = (G * D).sum(2, keepdims=True)
        c0 = (gi_ / r).sum(1, keepdims=True)
        dD = G * tau / r - gi_ * tau / (r * r) + c0
        # ---- D = s_prev K^T * scale
        ds = scale * np.einsum("bij,bjd->bid", dD, K) + dh_prev
        dK += scale * np.einsum("bij,bid->bjd", dD, s_prev)
    g["slots0"] = ds.sum(0)
    # ---- to_k MLP backward
    dz = dK
    g["to_k_w"] = np.zeros_like(W["to_k_w"])
    g["to_k_b"] = np.zeros_like(W["to_k_b"])
    for l in range(L - 1, -1, -1):
        if l < L - 1:
            dz = dz * (H[l + 1] > 0)
        g["to_k_w"][l] = np.einsum("bno,bni->oi", dz, H[l])
        g["to_k_b"][l] = dz.sum((0, 1))
        dz = dz @ W["to_k_w"][l]
    dX += dz            # X + PE
    g["dX"] = dX
    return g
