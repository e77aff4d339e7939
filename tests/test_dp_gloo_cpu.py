"""Multi-process data-parallel plumbing on CPU (gloo, world_size 2): construction-time broadcast, per-step flat
gradient all-reduce (mean), BatchNorm buffer broadcast before a training forward.  The arithmetic kernels are HIP
only, so the module under the wrapper is a stub that fills the gradient arena itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Stub(nn.Module):
    """Has the pieces scouter_amd.parallel relies on: parameters, BN buffers, _post_backward_hooks, a GradArena."""

    def __init__(self):
        super().__init__()
        from scouter_amd.nn_hip import BatchNorm2d, Conv2d
        self.conv = Conv2d(32, 64, 3, 1, 1)
        self.bn = BatchNorm2d(64)
        self._post_backward_hooks = []
        self._grad_ready_hooks = []
        self.arena = None

    def forward(self, fill):
        from scouter_amd.nn_hip import GradArena
        if self.arena is None:
            self.arena = GradArena(self)
        self.arena.flat.fill_(fill)            # "backward": every gradient element = fill
        half = self.arena.numel // 2 // 4 * 4
        for h in self._grad_ready_hooks:            # two buckets, tail first (like the real backward)
            h(self.arena, half, self.arena.numel)
        for h in self._grad_ready_hooks:
            h(self.arena, 0, half)
        self.arena.attach()
        for h in self._post_backward_hooks:
            h(self.arena)
        return self.arena


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scouter_amd.parallel import DistributedDataParallel
        torch.manual_seed(100 + rank)             # different initial weights per rank
        m = Stub()
        with torch.no_grad():
            m.bn.running_mean.fill_(float(rank + 1))
        w_before = m.conv.weight.detach().clone()
        ddp = DistributedDataParallel(m, device_ids=None)
        # 1. parameters / buffers equal rank 0's after construction
        gathered = [torch.empty_like(m.conv.weight.detach().contiguous()) for _ in range(world)]
        dist.all_gather(gathered, m.conv.weight.detach().contiguous())
        same_params = all(torch.equal(gathered[0], g) for g in gathered)
        # 2. gradient all-reduce = mean over ranks, delivered through param.grad views of the flat arena
        m.train()
        m.bn.running_mean.fill_(float(10 * (rank + 1)))       # diverge the buffer again
        arena = ddp(float(rank + 1))
        mean_ok = bool(torch.allclose(arena.flat, torch.full_like(arena.flat, (1 + world) / 2)))
        grad_view_ok = bool(torch.allclose(m.conv.weight.grad, torch.full_like(m.conv.weight, (1 + world) / 2))) and \
            m.conv.weight.grad.stride() == m.conv.weight.stride()
        # 3. buffers were re-broadcast from rank 0 before the training forward
        buf_ok = float(m.bn.running_mean[0]) == 10.0
        # 4. nothing is in flight between two forwards (round 4 prefetched the broadcast at the end of the backward
        #    and an eval forward / state_dict right after it raced with the collective): the forward consumed its own
        #    asynchronous broadcast -- the Stub has no backbone hook, so the wrapper's `finally` did
        buf_ok = buf_ok and ddp._buf_work is None
        # 5. train -> eval: the eval forward broadcasts too (like torch DDP), so every rank evaluates with rank 0's statistics
        m.bn.running_mean.fill_(float(7 * (rank + 1)))
        m.eval()
        with torch.no_grad():
            ddp(float(rank + 1))
        buf_ok = buf_ok and ddp._buf_work is None and float(m.bn.running_mean[0]) == 7.0
        m.train()
        dist.barrier()
        # 6. a buffer write on rank 0 ONLY (a rank-0-only load_state_dict): every rank issues the same collectives, nothing
        #    hangs, and DDP's "rank 0's buffers as of THIS forward" holds
        if rank == 0:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            sd["bn.running_mean"].fill_(100.0)
            m.load_state_dict(sd)
            import time
            time.sleep(0.2)                                   # (rank 0 late)
        ddp(float(rank + 1))
        buf_ok = buf_ok and float(m.bn.running_mean[0]) == 100.0 and float(m.bn.running_mean[-1]) == 100.0
        q.put((rank, same_params, mean_ok, grad_view_ok, buf_ok, bool(torch.equal(w_before, m.conv.weight)) == (rank == 0)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_wrapper_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_init_distributed_mode_env_contract(monkeypatch):
    """tools/prepare_things.py:9-31 contract: no RANK/WORLD_SIZE -> args.distributed = False, nothing initialised."""
    import argparse
    from scouter_amd.tools import prepare_things as prt
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLURM_PROCID"):
        monkeypatch.delenv(k, raising=False)
    args = argparse.Namespace(dist_url="env://", device="cpu")
    prt.init_distributed_mode(args)
    assert args.distributed is False
    assert prt.get_world_size() == 1 and prt.get_rank() == 0 and prt.is_main_process()


def test_train_arg_parser_matches_reference_defaults():
    """Defaults read from the reference's train.py:28-78."""
    from scouter_amd.train import get_args_parser, checkpoint_name
    a = get_args_parser().parse_args([])
    assert (a.model, a.dataset, a.channel, a.lr, a.lr_drop, a.batch_size, a.epochs) == \
        ("resnet18", "MNIST", 512, 1e-4, 70, 64, 10)
    assert (a.num_classes, a.slots_per_class, a.power, a.lambda_value, a.hidden_dim, a.to_k_layer, a.freeze_layers,
            a.loss_status, a.img_size) == ("10", "3", "2", "1.", 64, 1, 2, 1, 260)
    assert a.pre_trained is True and a.use_slot is True and a.vis is False and a.device == "cuda"
    a.cal_area_size = False
    assert checkpoint_name(a) == "MNIST_use_slot_checkpoint.pth"
    a.loss_status = -1
    assert checkpoint_name(a, "checkpoint0009.pth") == "MNIST_use_slot_negative_checkpoint0009.pth"
