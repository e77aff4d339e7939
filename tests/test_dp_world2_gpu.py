"""World-size-2 data-parallel step of the REAL model on one GPU (SURVEY section 8 rows a17 / e; reference
train.py:139-141 DDP wrap, :152-154 DistributedSampler sharding, tools/prepare_things.py:9-31).

Two spawned processes share cuda:0 and talk over `gloo` on device tensors (RCCL refuses two ranks on one device; the
driver measures the RCCL path at N > 1).  Everything above the backend is the production code: SlotModel's backward
with its per-stage `stage_done` bucket hooks, parallel._launch_bucket (async all-reduce per bucket while earlier layers
are still in backward), _finish_gradients (wait + 1/world scale), the construction broadcast of the permuted-layout
conv weights, the per-step BatchNorm-buffer broadcast, FusedAdamW on the reduced arena.

Equivalence that must hold (what DDP guarantees in the reference): with per-sample-independent layers -- BatchNorm in
eval mode, the head in train mode -- and a loss that is a batch mean (power = 1), the rank-mean of the shard gradients
equals the full-batch gradient of one process; checked against this build's own single-process run AND against the
oracle's fp64 autograd.  (Train-mode BatchNorm uses per-GPU statistics in the reference too -- no SyncBN -- so shards
are not equivalent to the full batch there by construction: SURVEY section 4 (iii).)"""
import argparse
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORLD, BATCH, IMG = 2, 8, 64


def _args():
    return argparse.Namespace(model="resnet18", pre_trained=False, num_classes=10, dataset="MNIST", use_slot=True,
                              use_pre=False, grad=False, channel=512, slots_per_class=1, hidden_dim=64,
                              freeze_layers=0, vis=False, vis_id=0, loss_status=1, power=1, to_k_layer=1,
                              lambda_value="1")


def _build(device="cuda"):
    """SlotModel on the synthetic reference-layout state; BatchNorm in eval mode, everything else training."""
    from oracle import torch_oracle as O
    from scouter_amd.nn_hip import BatchNorm2d
    from scouter_amd.sloter.slot_model import SlotModel
    spec = O.state_dict_spec("resnet18", 10, 1, 1, in_chans=1, mnist_stem=True)
    P = O.synth_state(spec, 300)
    m = SlotModel(_args())
    m.load_state_dict(P)
    m = m.to(device).train()
    for mod in m.modules():
        if isinstance(mod, BatchNorm2d):
            mod.eval()
    return m, P


def _one_step(net, model, images, labels, lr=1e-3):
    from scouter_amd.optim import FusedAdamW
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=lr)
    opt.zero_grad()
    out, losses = net(images, labels)
    losses[0].backward()
    torch.cuda.synchronize()
    grads = model.grad_arena().flat.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return out.detach().cpu(), float(losses[0]), grads.cpu(), params


def _worker(rank, port, outdir, backend="gloo"):
    import torch.distributed as dist
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(rank if backend == "nccl" else 0)          # RCCL: one rank per device; gloo: both on cuda:0
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    try:
        from oracle import torch_oracle as O
        from scouter_amd.parallel import DistributedDataParallel
        model, _ = _build()
        if rank != 0:                      # diverge rank 1: the construction broadcast must bring rank 0's state back
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.5)
                for b in model.buffers():
                    if b.dtype.is_floating_point:
                        b.add_(0.25)
        net = DistributedDataParallel(model, device_ids=[torch.cuda.current_device()], find_unused_parameters=True)
        after_ctor = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        if rank != 0:                      # diverge the running statistics again: re-broadcast before each forward
            with torch.no_grad():
                model.backbone.bn1.running_mean.add_(3.0)
                model.backbone.layer4[1].bn2.running_var.mul_(2.0)
        images, labels = O.synth_batch(BATCH, 1, IMG, 10, 310)
        lo, hi = rank * BATCH // WORLD, (rank + 1) * BATCH // WORLD          # DistributedSampler-style shard
        out, loss, grads, params = _one_step(net, model, images[lo:hi].cuda(), labels[lo:hi].cuda())
        assert net._pending == []
        torch.save(dict(after_ctor=after_ctor, out=out, loss=loss, grads=grads, params=params,
                        buckets=getattr(net, "_buckets_launched", None)), os.path.join(outdir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_world2_real_backward_matches_single_process_and_oracle(tmp_path):
    import torch.multiprocessing as mp
    from oracle import torch_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "rank process failed (exit code %s)" % p.exitcode
    r0, r1 = (torch.load(tmp_path / ("rank%d.pt" % r), weights_only=False) for r in range(WORLD))

    # 1. construction broadcast: rank 1 holds rank 0's parameters and buffers (bit-exact, permuted conv layouts incl.)
    model, P = _build()
    for k, v in r0["after_ctor"].items():
        assert torch.equal(v, P[k]), k
        assert torch.equal(r1["after_ctor"][k], v), k

    # 2. both ranks hold the same reduced gradients and, after FusedAdamW, the same parameters; buffers of rank 1 were
    #    re-broadcast from rank 0 before the forward (eval-mode BatchNorm: they are unchanged by the step)
    assert torch.equal(r0["grads"], r1["grads"])
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        if k.endswith(("running_mean", "running_var")):
            assert torch.equal(r0["params"][k], P[k]), k

    # 3. rank-mean of the shard gradients == the full-batch gradient of ONE process (same kernels, other summation
    #    order over the batch -> fp32 round-off), and the stepped parameters agree
    images, labels = O.synth_batch(BATCH, 1, IMG, 10, 310)
    out, loss, grads, params = _one_step(model, model, images.cuda(), labels.cuda())
    arena = model.grad_arena()
    np.testing.assert_allclose(0.5 * (r0["loss"] + r1["loss"]), loss, rtol=2e-6)
    np.testing.assert_allclose(torch.cat([r0["out"], r1["out"]]).numpy(), out.numpy(), atol=2e-6)
    for name, p, off, n in arena.entries:
        a, b = r0["grads"][off:off + n], grads[off:off + n]
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
    for k in params:
        if params[k].dtype.is_floating_point:
            # AdamW's first step moves every weight by ~lr*sign(g): elements whose gradient is ~0 may flip, so compare
            # through the update size
            assert float((r0["params"][k] - params[k]).abs().max()) <= 2.5e-3, k
    moved = sum(float((params[k] - P[k]).abs().max()) > 0 for k in params if params[k].dtype.is_floating_point)
    assert moved > 50

    # 4. against the oracle: fp64 autograd of the full batch (eval-mode BatchNorm), every trainable tensor
    Pd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
    keys = O.trainable_keys(Pd)
    leaves = {k: Pd[k].detach().requires_grad_(True) for k in keys}
    Q = dict(Pd); Q.update(leaves)
    cfg = dict(model="resnet18", num_classes=10, slots_per_class=1, loss_status=1, power=1, lambda_value=1.0)
    _, ref_losses = O.slot_model_forward(Q, images.double(), labels, cfg, training=False)
    ref = dict(zip(keys, torch.autograd.grad(ref_losses[0], [leaves[k] for k in keys], allow_unused=True)))
    np.testing.assert_allclose(0.5 * (r0["loss"] + r1["loss"]), float(ref_losses[0]), rtol=1e-5)
    checked = 0
    for name, p, off, n in arena.entries:
        g = ref.get(name)
        if g is None:
            continue
        mine = r0["grads"][off:off + n]
        mine = mine.view(p.shape[2], p.shape[3], p.shape[1], p.shape[0]).permute(3, 2, 0, 1) if p.dim() == 4 \
            else mine.view(p.shape)
        scale = max(float(g.abs().max()), 1e-7)
        err = float((mine.double() - g).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked >= 60


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices: RCCL refuses two ranks on one device")
def test_world2_over_rccl_on_two_devices(tmp_path):
    """The same world-2 step over the PRODUCTION backend (torch.distributed 'nccl' == RCCL over xGMI), one rank per device
    (the single-GPU lease skips it): both ranks end with bit-identical reduced gradients and parameters,
    the rank-mean gradient equals one process's full-batch gradient, and `bench.py --gpus 2` launches its own two ranks and
    reports the group size RCCL ran with.  Reference: train.py:139-141 (DDP over the launcher's ranks)."""
    import json
    import subprocess
    import sys
    import torch.multiprocessing as mp
    from oracle import torch_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, port, str(tmp_path), "nccl")) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "rank process failed (exit code %s)" % p.exitcode
    r0, r1 = (torch.load(tmp_path / ("rank%d.pt" % r), weights_only=False) for r in range(WORLD))
    assert torch.equal(r0["grads"], r1["grads"])
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    model, P = _build()
    images, labels = O.synth_batch(BATCH, 1, IMG, 10, 310)
    out, loss, grads, params = _one_step(model, model, images.cuda(), labels.cuda())
    np.testing.assert_allclose(0.5 * (r0["loss"] + r1["loss"]), loss, rtol=2e-6)
    for name, p, off, n in model.grad_arena().entries:
        a, b = r0["grads"][off:off + n], grads[off:off + n]
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "1", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-prof"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["data_parallel"]["rccl_ranks"] == 2 and line["data_parallel"]["backend"] == "nccl"
    assert line["data_parallel"]["buckets"] == 5 and line["scaling"] == "weak"


def test_world2_buckets_are_launched_per_stage_during_backward():
    """The real backward hands FIVE gradient ranges to the wrapper (head + layer4, layer3, layer2, layer1, stem), last
    arena range first, covering the arena exactly once -- recorded on a 1-rank gloo group over device tensors."""
    import torch.distributed as dist
    from oracle import torch_oracle as O
    from scouter_amd.parallel import DistributedDataParallel
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        model, _ = _build()
        net = DistributedDataParallel(model, device_ids=[0])
        seen = []
        orig = net._launch_bucket
        idx = model._grad_ready_hooks.index(orig)
        model._grad_ready_hooks[idx] = lambda arena, lo, hi: (seen.append((lo, hi)), orig(arena, lo, hi))[1]
        images, labels = O.synth_batch(4, 1, IMG, 10, 311)
        out, losses = net(images.cuda(), labels.cuda())
        losses[0].backward()
        torch.cuda.synchronize()
        n = model.grad_arena().numel
        assert len(seen) == 5 and seen[0][1] == n and seen[-1][0] == 0
        assert all(seen[i][0] == seen[i + 1][1] for i in range(4)) and all(lo < hi for lo, hi in seen)
    finally:
        dist.destroy_process_group()
