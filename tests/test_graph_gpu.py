"""hipGraph capture of the training step (scouter_amd/graph.py): replays must be bit-identical to the eager step --
same kernels, same order, same arithmetic -- including the device-side AdamW step counter and a learning-rate change
between replays."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed=5, arch="resnest26d", size=64, B=4):
    from oracle import torch_oracle as O
    from scouter_amd.sloter.slot_model import SlotModel
    args = argparse.Namespace(model=arch, pre_trained=False, num_classes=10, dataset="ImageNet", use_slot=True,
                              use_pre=False, grad=False, channel=O.ARCHS[arch]["channel"], slots_per_class=1,
                              hidden_dim=64, freeze_layers=0, vis=False, vis_id=0, loss_status=1, power=2,
                              to_k_layer=3, lambda_value="1")
    P = O.synth_state(O.state_dict_spec(arch, 10, 1, 3), seed)
    m = SlotModel(args)
    m.load_state_dict(P)
    batches = [O.synth_batch(B, 3, size, 10, 100 + i) for i in range(4)]
    return m.cuda().train(), [(x.cuda(), y.cuda()) for x, y in batches]


def test_graph_replay_is_bit_identical_to_the_eager_step():
    from scouter_amd.graph import GraphedTrainStep
    from scouter_amd.optim import FusedAdamW
    # eager reference: warm-up steps on batch 0 (what capture does), then one step per batch, lr changed before the last
    m_e, batches = _model()
    opt_e = FusedAdamW([p for p in m_e.parameters() if p.requires_grad], lr=1e-3)
    warm = 3
    outs_e = []
    seq = [batches[0]] * warm + batches
    for i, (x, y) in enumerate(seq):
        if i == len(seq) - 1:
            opt_e.param_groups[0]["lr"] = 2.5e-4
        opt_e.zero_grad()
        out, losses = m_e(x, y)
        losses[0].backward()
        opt_e.step()
        outs_e.append((out.detach().clone(), m_e.last_stats.clone()))
    torch.cuda.synchronize()

    m_g, _ = _model()
    opt_g = FusedAdamW([p for p in m_g.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    step = GraphedTrainStep(m_g, opt_g, warmup=warm, preserve_state=False).prepare(*batches[0])
    for i, (x, y) in enumerate(batches):
        if i == len(batches) - 1:
            opt_g.param_groups[0]["lr"] = 2.5e-4
        logp, stats = step(x, y)
        ref_out, ref_stats = outs_e[warm + i]
        assert torch.equal(logp, ref_out), i
        assert torch.equal(stats[:5], ref_stats[:5]), i
    torch.cuda.synchronize()
    assert step.replays == len(batches)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt_g.state["_flat_0"]["step"] == opt_e.state["_flat_0"]["step"] == warm + len(batches)
    assert torch.equal(opt_g.state["_flat_0"]["exp_avg"], opt_e.state["_flat_0"]["exp_avg"])
    assert torch.equal(opt_g.state["_flat_0"]["exp_avg_sq"], opt_e.state["_flat_0"]["exp_avg_sq"])
    with pytest.raises(RuntimeError, match="static shapes"):
        step(batches[0][0][:2], batches[0][1][:2])


def test_graphed_training_is_step_for_step_the_eager_training():
    """preserve_state (the default): the warm-up executions capture needs leave no trace -- parameters, BatchNorm running
    statistics (incl. num_batches_tracked) and the optimizer state after N graphed steps equal N eager steps bit for bit."""
    from scouter_amd.graph import GraphedTrainStep
    from scouter_amd.optim import FusedAdamW
    m_e, batches = _model(seed=9, arch="resnet18", size=96, B=6) if False else _model(seed=9)
    opt_e = FusedAdamW([p for p in m_e.parameters() if p.requires_grad], lr=1e-3)
    stats_e = []
    for x, y in batches:
        opt_e.zero_grad()
        out, losses = m_e(x, y)
        losses[0].backward()
        opt_e.step()
        stats_e.append(m_e.last_stats.clone())
    m_g, _ = _model(seed=9)
    opt_g = FusedAdamW([p for p in m_g.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    step = GraphedTrainStep(m_g, opt_g)
    for i, (x, y) in enumerate(batches):
        logp, stats = step(x, y)
        assert torch.equal(stats[:5], stats_e[i][:5]), i
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt_g.state["_flat_0"]["step"] == opt_e.state["_flat_0"]["step"] == len(batches)
    assert torch.equal(opt_g.state["_flat_0"]["exp_avg_sq"], opt_e.state["_flat_0"]["exp_avg_sq"])


def test_capturable_adamw_matches_host_stepped_adamw_eagerly():
    from scouter_amd.optim import FusedAdamW
    torch.manual_seed(0)
    outs = []
    for cap in (False, True):
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.nn.Parameter(torch.randn(70001, device="cuda", generator=g))
        flat = torch.zeros(70004, device="cuda")
        p.grad = flat[:70001]
        opt = FusedAdamW([p], lr=3e-3, capturable=cap)
        for t in range(5):
            flat.copy_(torch.randn(70004, device="cuda", generator=g))
            if t == 3:
                opt.param_groups[0]["lr"] = 1e-3
                opt.sync_hyperparameters()
            opt.step()
        outs.append((p.detach().clone(), opt.state["_flat_0"]["step"]))
    assert outs[0][1] == outs[1][1] == 5
    # the bias corrections are formed in fp64 on the host vs in the kernel from the same fp64 formula
    assert torch.allclose(outs[0][0], outs[1][0], rtol=0, atol=1e-7)


def test_eval_forward_and_empty_cache_between_replays_leave_the_graph_intact():
    """the captured graph has the plane-weight pointer table and plane buffers of the TRAINING key baked in;
    an eval forward in between (forward planes only: another key) must not free or replace them, even when the caching
    allocator is emptied and refilled with other tensors before the next replay."""
    from scouter_amd.graph import GraphedTrainStep
    from scouter_amd.optim import FusedAdamW
    m_e, batches = _model(seed=9)
    opt_e = FusedAdamW([p for p in m_e.parameters() if p.requires_grad], lr=1e-3)
    warm = 2
    for x, y in [batches[0]] * warm + batches:
        opt_e.zero_grad()
        out, losses = m_e(x, y)
        losses[0].backward()
        opt_e.step()
    torch.cuda.synchronize()

    m_g, _ = _model(seed=9)
    opt_g = FusedAdamW([p for p in m_g.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    step = GraphedTrainStep(m_g, opt_g, warmup=warm, preserve_state=False).prepare(*batches[0])
    n_keys = len(m_g._wsplitter._entries)
    for i, (x, y) in enumerate(batches):
        step(x, y)
        if i < len(batches) - 1:
            m_g.eval()
            with torch.no_grad():
                m_g(x, y)                                   # forward planes only: a second splitter key
            m_g.train()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(16)]   # whoever gets freed blocks
            del junk
    torch.cuda.synchronize()
    assert len(m_g._wsplitter._entries) == n_keys + 1      # the eval key was ADDED, the training key kept
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
