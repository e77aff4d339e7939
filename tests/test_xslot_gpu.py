"""Parity of the fused xSlot head (conv1x1+ReLU -> PE -> xslot fwd/bwd -> loss) on the GPU against
 (a) the golden vectors generated from the reference (tests/golden/head_*.npz) and
 (b) the CPU oracle (oracle/torch_oracle.py) evaluated on the same seeded inputs, incl. its fp64 "truth".
Tolerance: north_star asks 1e-4 fp32 on logits / attention; for S >= 200 the reference's own fp32-vs-fp64 spread is
of that order, so the bound is max(1e-4, 3 x that spread); in practice the HIP path is closer to
the fp64 truth than the reference's fp32 run is, which is asserted too (with slack)."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as O                       # noqa: E402
from oracle.gen_golden import HEAD_CASES, LAMBDA, head_inputs   # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_head_model(C, spc, L, ls, power, Cin):
    from scouter_amd.sloter.slot_model import SlotModel
    args = argparse.Namespace(model="resnet18", pre_trained=False, num_classes=C, dataset="MNIST", use_slot=True,
                              use_pre=False, grad=False, channel=Cin, slots_per_class=spc, hidden_dim=64,
                              freeze_layers=0, vis=False, vis_id=0, loss_status=ls, power=power, to_k_layer=L,
                              lambda_value=LAMBDA)
    return SlotModel(args)


def run_head(case):
    C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
    feat, labels, P = head_inputs(case)
    m = make_head_model(C, spc, L, ls, power, Cin)
    sd = m.state_dict()
    sd.update({k: v for k, v in P.items()})
    m.load_state_dict(sd)
    m = m.cuda()
    m.grad_arena()
    fd = feat.permute(0, 2, 3, 1).contiguous().cuda()
    logp, stats, hstate = m._head_forward(fd, labels.cuda(), True)
    one = torch.ones((), device="cuda")
    dfeat = m._head_backward(hstate, None, one, None, None, True)
    m.grad_arena().attach()
    torch.cuda.synchronize()
    return m, logp, stats, hstate, dfeat.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("case", list(HEAD_CASES))
def test_head_matches_reference_golden_and_oracle(case):
    C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
    g = np.load(os.path.join(GOLD, "head_%s.npz" % case))
    m, logp, stats, hstate, dfeat = run_head(case)
    so = hstate[3]
    floor = float(np.abs(g["f32_logits"] - g["f64_logits"]).max())
    tol = max(1e-4, 3 * floor)
    logits = so["logits"].cpu().numpy()
    err_hip = np.abs(logits - g["f64_logits"]).max()
    assert err_hip <= tol, (err_hip, tol)
    assert err_hip <= max(3 * floor, 2e-5), "HIP logits should sit inside the reference's own fp32 noise"
    np.testing.assert_allclose(logits, g["f32_logits"], atol=tol, rtol=0)
    np.testing.assert_allclose(logp.cpu().numpy(), g["f32_log_probs"], atol=tol, rtol=0)
    np.testing.assert_allclose(so["attn"].cpu().numpy(), g["f64_attn"], atol=max(1e-4, tol), rtol=0)
    np.testing.assert_allclose(stats[:3].cpu().numpy(), [float(g["f64_loss"]), float(g["f64_nll"]), float(g["f64_area"])],
                               atol=tol, rtol=1e-5)
    vis = m.slot.vis_maps()
    ref_map = g["f64_attn"][0]
    if C * spc == ref_map.shape[0] and float(ref_map.max() - ref_map.min()) < 1e-9:
        # single-token grid: D / r_i == 1, every slot's attention is sigmoid(tau) and the reference's min-max
        # normalisation is 0/0 (its uint8 image is a cast of NaN); the meaningful statement is "the map is constant"
        a0 = so["attn"][0].cpu().numpy()
        assert float(a0.max() - a0.min()) <= 1e-6
    else:
        assert np.abs(vis.astype(int) - g["f32_vis"].astype(int)).max() <= 1
    # ---- gradients against the oracle's fp64 autograd on the same inputs
    feat, labels, P = head_inputs(case)
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    f64 = feat.double().requires_grad_(True)
    cfg = dict(num_classes=C, slots_per_class=spc, loss_status=ls, power=power, lambda_value=float(LAMBDA))
    out, losses = O.head_forward(leaves, f64, labels, cfg)
    losses[0].backward()
    gfloor = 30 * floor

    def chk(mine, ref, name):
        ref = ref.numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        np.testing.assert_allclose(mine.detach().cpu().double().numpy(), ref, atol=2e-3 * scale + gfloor, rtol=2e-3,
                                   err_msg=name)
    chk(dfeat, f64.grad, "dfeat")
    named = dict(m.named_parameters())
    for k in leaves:
        if "to_q" in k:
            assert named[k].grad is None
            continue
        chk(named[k].grad, leaves[k].grad, k)
    np.testing.assert_allclose(dfeat[:, :16].cpu().numpy(), g["f32_dfeat_head"],
                               atol=2e-3 * np.abs(g["f32_dfeat_head"]).max() + gfloor, rtol=5e-3)


def test_head_fwd_is_deterministic_and_ragged_batch():
    """Same inputs -> bit-identical outputs; batch entries are independent (ragged last block)."""
    case = "c2_in10_pos"
    m, logp, stats, hstate, dfeat = run_head(case)
    m2, logp2, stats2, _, dfeat2 = run_head(case)
    assert torch.equal(logp, logp2) and torch.equal(dfeat, dfeat2) and torch.equal(stats[:3], stats2[:3])


def test_xslot_large_batch_properties():
    """BASELINE-size head (B=256, S=300, N=49): size-independent properties -- permutation equivariance over the
    batch, attention in (0,1), logits = loss_status/d * sum_j A[s,j] * rowsum(X)[j]."""
    from scouter_amd import kernels as Kk
    rng = np.random.default_rng(12)
    B, N, d, C, spc, L = 256, 49, 64, 100, 3, 3
    S = C * spc
    spec = {k: v for k, v in O.state_dict_spec("resnet18", C, spc, L).items() if k.startswith("slot.")}
    P = {k: v.cuda() for k, v in O.synth_state(spec, 5).items()}
    X = torch.from_numpy(np.maximum(rng.standard_normal((B, N, d)), 0).astype(np.float32)).cuda()
    PE = Kk.posenc_sine(7, 7, d, X.device)
    tw = [P["slot.to_k.%d.weight" % (2 * l)] for l in range(L)]
    tb = [P["slot.to_k.%d.bias" % (2 * l)] for l in range(L)]
    args = (P["slot.initial_slots"][0].contiguous(), P["slot.gru.weight_ih_l0"], P["slot.gru.weight_hh_l0"],
            P["slot.gru.bias_ih_l0"], P["slot.gru.bias_hh_l0"], spc, 3, 1.0)
    o1 = Kk.xslot_fwd(X, PE, tw, tb, *args)
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    o2 = Kk.xslot_fwd(X[perm].contiguous(), PE, tw, tb, *args)
    assert torch.equal(o1["logits"][perm], o2["logits"]) and torch.equal(o1["attn"][perm], o2["attn"])
    A = o1["attn"]
    assert float(A.min()) >= 0.0 and float(A.max()) <= 1.0 and torch.isfinite(o1["logits"]).all()
    xs = X.double().sum(2)                                                  # [B, N]
    lg = (A.double() * xs[:, None, :]).sum(2).view(B, C, spc).sum(2) / d
    np.testing.assert_allclose(o1["logits"].cpu().double().numpy(), lg.cpu().numpy(), atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(o1["area_part"].cpu().double().numpy(), A.double().sum((1, 2)).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("B,S,N,T,L,spc", [(2, 512, 64, 8, 2, 1),      # S, T at the kernel limits, 16 slot tiles
                                            (2, 512, 96, 3, 8, 2),      # N, L at the limits (scratch variant)
                                            (3, 33, 20, 1, 1, 3),       # a single iteration (no GRU), ragged tiles
                                            (1, 96, 32, 2, 1, 1),       # one image, N = one token tile exactly
                                            # leftover slot tiles walked as 16-slot half tiles by the backward (round 6):
                                            (2, 289, 64, 2, 1, 1),      # 8 tiles + 3 halves, the last with ONE live slot
                                            (2, 160, 25, 3, 2, 1),      # 4 tiles + 2 halves, one token tile
                                            (2, 40, 33, 3, 1, 2)])      # no whole tile at all: 3 halves
def test_xslot_kernels_at_their_limits_match_the_oracle(B, S, N, T, L, spc):
    """The fused forward / backward kernels straight through the C ABI at the supported maxima (S <= 512 slots, N <= 96
    tokens, T <= 8 iterations, L <= 8 to_k layers) and at degenerate sizes, against the fp64 oracle and its autograd.
    Yardstick as everywhere for many-slot heads: what plain fp32 PyTorch loses on the same inputs."""
    from scouter_amd import kernels as Kk
    d, C = 64, S // spc
    g = torch.Generator().manual_seed(S * 7 + N)
    r = lambda *sh: torch.randn(*sh, generator=g)
    P = {"slot.initial_slots": (r(1, S, d).abs() * 0.5), "slot.gru.weight_ih_l0": r(3 * d, d) * 0.1,
         "slot.gru.weight_hh_l0": r(3 * d, d) * 0.1, "slot.gru.bias_ih_l0": r(3 * d) * 0.1,
         "slot.gru.bias_hh_l0": r(3 * d) * 0.1}
    for l in range(L):
        P["slot.to_k.%d.weight" % (2 * l)] = r(d, d) * (0.2 if L > 2 else 0.1)
        P["slot.to_k.%d.bias" % (2 * l)] = r(d) * 0.1
    X, PE = r(B, N, d).relu(), r(N, d) * 0.3
    wl, ga = r(B, C), 0.01

    def oracle(dtype):
        Q = {k: v.to(dtype).clone().requires_grad_(True) for k, v in P.items()}
        x = X.to(dtype).clone().requires_grad_(True)
        aux = {}
        lg, _ = O.xslot_forward(Q, x + PE.to(dtype), x, C, spc, 1, 1, iters=T, aux=aux)
        ((lg * wl.to(dtype)).sum() + ga * aux["attn"].sum()).backward()
        return lg.detach(), aux["attn"].detach(), x.grad, Q

    lg64, at64, gx64, Q64 = oracle(torch.float64)
    lg32, at32, gx32, Q32 = oracle(torch.float32)
    floor = float((lg32.double() - lg64).abs().max())
    tol = max(1e-4, 3 * floor)
    dev = torch.device("cuda")
    cu = lambda t: t.float().contiguous().to(dev)
    tw = [cu(P["slot.to_k.%d.weight" % (2 * l)]) for l in range(L)]
    tb = [cu(P["slot.to_k.%d.bias" % (2 * l)]) for l in range(L)]
    gru = [cu(P["slot.gru." + n]) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    s0 = cu(P["slot.initial_slots"][0])
    fwd = Kk.xslot_fwd(cu(X), cu(PE), tw, tb, s0, *gru, spc, T, 1)
    np.testing.assert_allclose(fwd["logits"].cpu().numpy(), lg64.numpy(), atol=tol, rtol=1e-4)
    np.testing.assert_allclose(fwd["attn"].cpu().numpy(), at64.numpy(), atol=max(tol, 1e-4), rtol=0)
    bwd = Kk.xslot_bwd(cu(X), cu(PE), tw, s0, *gru, fwd, cu(wl), torch.full((1,), ga, device=dev), spc, T, 1)
    torch.cuda.synchronize()

    def close(mine, ref, ref32, name):
        sc = float(ref.abs().max())
        noise = float((ref32.double() - ref).abs().max())                      # fp32 autograd's own deviation
        assert float((mine.cpu().double() - ref).abs().max()) <= 3e-3 * sc + 3 * noise + 1e-6, name

    close(bwd["dX"], gx64, gx32, "dX")
    close(bwd["ds0"].sum(0), Q64["slot.initial_slots"].grad[0], Q32["slot.initial_slots"].grad[0], "d initial_slots")
    if T > 1:       # GRU weight gradients are GEMMs over the rows the kernel emits: dW_ih = dgi^T U, db_ih = colsum(dgi)
        dgi, U = bwd["dgi"].double().cpu().reshape(-1, 3 * d), bwd["U"].double().cpu().reshape(-1, d)
        close(dgi.t() @ U, Q64["slot.gru.weight_ih_l0"].grad, Q32["slot.gru.weight_ih_l0"].grad, "dW_ih")
        close(bwd["dgh"].double().cpu().reshape(-1, 3 * d).sum(0), Q64["slot.gru.bias_hh_l0"].grad,
              Q32["slot.gru.bias_hh_l0"].grad, "db_hh")
    # first to_k layer: dW_0 = dZ_0^T (X + PE), db_0 = colsum(dZ_0)
    dz0 = bwd["dZ"][0].double().cpu().reshape(-1, d)
    close(dz0.t() @ (X + PE).double().reshape(-1, d), Q64["slot.to_k.0.weight"].grad, Q32["slot.to_k.0.weight"].grad, "dW_to_k0")


@pytest.mark.parametrize("B,S,N,T", [(2, 300, 49, 3), (2, 289, 64, 2), (2, 160, 25, 3), (2, 40, 33, 3), (2, 21, 9, 3)])
def test_xslot_backward_half_tiles_agree_with_whole_tiles(B, S, N, T, monkeypatch):
    """ntiles % 4 = 1, 2: the leftover 32-slot tiles are walked as 16-slot half tiles, one per wave, on 16x16x4 MFMAs
    (xslot_bwd.hip, `halves`); SCOUTER_XSLOT_BWD_HALVES=0 walks them as whole tiles.  Same maths, different summation order:
    the two launches agree to rounding on every output."""
    from scouter_amd import kernels as Kk
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(B * 1000 + S)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    d, L, spc = 64, 2, 1
    X, PE = r(B, N, d).relu_(), r(N, d) * 0.3
    tw, tb = [r(d, d) * 0.1 for _ in range(L)], [r(d) * 0.1 for _ in range(L)]
    s0 = r(S, d).abs() * 0.5
    gru = (r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1)
    fwd = Kk.xslot_fwd(X, PE, tw, tb, s0, *gru, spc, T, 1)
    dl, ga = r(B, S // spc), torch.full((1,), 0.01, device=dev)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SCOUTER_XSLOT_BWD_HALVES", flag)
        o = Kk.xslot_bwd(X, PE, tw, s0, *gru, fwd, dl, ga, spc, T, 1)
        torch.cuda.synchronize()
        outs[flag] = {k: v.clone() for k, v in o.items()}
    for k in outs["1"]:
        a, b_ = outs["1"][k].double(), outs["0"][k].double()
        assert torch.isfinite(a).all(), k
        assert float((a - b_).abs().max()) <= 2e-4 * float(b_.abs().max()) + 1e-7, k


@pytest.mark.parametrize("B,S,N,T", [(3, 300, 49, 3), (2, 64, 49, 2), (3, 160, 25, 3), (2, 96, 81, 3)])
def test_xslot_backward_is_reproducible_and_ignores_stale_workspace(B, S, N, T):
    """Several slot tiles per image (S > 32: the four waves of a workgroup exchange tau / c0 / partial sums): launches
    on the same inputs are bit-identical, whatever the shared workspace and the recycled output buffers held before
    (zeros, NaNs, the previous launch's data) -- i.e. nothing is read before it is written and no hand-off races."""
    from scouter_amd import kernels as Kk
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(B * 1000 + S)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    d, L, spc = 64, 2, 1
    X, PE = r(B, N, d).relu_(), r(N, d) * 0.3
    tw, tb = [r(d, d) * 0.1 for _ in range(L)], [r(d) * 0.1 for _ in range(L)]
    s0 = r(S, d).abs() * 0.5
    gru = (r(3 * d, d) * 0.1, r(3 * d, d) * 0.1, r(3 * d) * 0.1, r(3 * d) * 0.1)
    fwd = Kk.xslot_fwd(X, PE, tw, tb, s0, *gru, spc, T, 1)
    dl, ga = r(B, S // spc), torch.full((1,), 0.01, device=dev)
    outs = []
    for i, fill in enumerate([0.0, None, float("nan"), None, None, 7.0, None, None]):
        if fill is not None:
            Kk.workspace(1, dev).view(torch.float32).fill_(fill)
            junk = torch.empty(32 << 20, device=dev).fill_(fill)            # what torch.empty hands out next
            del junk
        o = Kk.xslot_bwd(X, PE, tw, s0, *gru, fwd, dl, ga, spc, T, 1)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items()})
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all(), k
        for i, o in enumerate(outs[1:], 1):
            assert torch.equal(outs[0][k], o[k]), (k, i)


@pytest.mark.parametrize("case", ["c2_in10_pos", "c3_in10_neg", "grid9_spc3", "c5_in100_spc3"])
def test_standalone_slot_attention_forward_is_differentiable(case):
    """The reference's module API, `SlotAttention(...)(x + pe, x) -> (logits, area ** power)`, used outside SlotModel:
    values and every gradient (x and all parameters but the unused to_q) against the oracle's fp64 autograd."""
    from scouter_amd.sloter.utils.slot_attention import SlotAttention
    C, spc, side, L, ls, power, B, Cin = HEAD_CASES[case]
    feat, labels, P = head_inputs(case)
    rng = np.random.default_rng(5)
    N, d = side * side, 64
    x = torch.from_numpy(rng.standard_normal((B, N, d)).clip(0)).float()            # post-ReLU tokens
    pe = torch.from_numpy(rng.standard_normal((N, d)) * 0.3).float()
    mod = SlotAttention(C, spc, d, loss_status=ls, power=power, to_k_layer=L)
    sd = mod.state_dict()
    sd.update({k[len("slot."):]: v for k, v in P.items() if k.startswith("slot.")})
    mod.load_state_dict(sd)
    mod = mod.cuda()
    xd = x.cuda().requires_grad_(True)
    logits, term = mod(xd + pe.cuda(), xd)
    wl = torch.from_numpy(rng.standard_normal((B, C))).float()
    (logits * wl.cuda()).sum().add(term * 3.0).backward()
    torch.cuda.synchronize()
    # oracle, fp64
    Pd = {k: v.double().clone().requires_grad_(True) for k, v in P.items() if k.startswith("slot.")}
    x64 = x.double().requires_grad_(True)
    rl, rt = O.xslot_forward(Pd, x64 + pe.double(), x64, C, spc, ls, power)
    ((rl * wl.double()).sum() + rt * 3.0).backward()
    floor = 1e-4 if C * spc < 100 else 2e-3                                          # S >= 200: ill-conditioned (fact 10)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), rl.detach().numpy(), atol=floor, rtol=1e-4)
    np.testing.assert_allclose(float(term.detach()), float(rt.detach()), rtol=1e-4, atol=1e-6)
    gx = xd.grad.cpu().double()
    assert float((gx - x64.grad).abs().max()) <= 2e-3 * float(x64.grad.abs().max()) + 10 * floor
    for name, p in mod.named_parameters():
        ref = Pd["slot." + name].grad
        if name.startswith("to_q"):
            assert p.grad is None and ref is None
            continue
        sc = float(ref.abs().max())
        assert float((p.grad.cpu().double() - ref).abs().max()) <= 3e-3 * sc + 30 * floor, name
