"""Data-parallel training over RCCL/xGMI -- the MI355X counterpart of the reference's
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)` (train.py:139-141).

One process per GPU (torch.distributed, backend 'nccl' == RCCL on ROCm).  Because every parameter gradient of the HIP
path lives in ONE flat fp32 arena, the per-step exchange is a single sum-all-reduce of that buffer (43-98 MiB)
followed by a scale by 1/world -- no bucketing machinery, no graph walk for unused parameters (`to_q` is excluded
statically).  Like DDP: parameters and buffers are broadcast from rank 0 at construction, and BatchNorm running
statistics are re-broadcast from rank 0 before each training forward (`broadcast_buffers=True`)."""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import kernels as K


class DistributedDataParallel(nn.Module):
    def __init__(self, module, device_ids=None, find_unused_parameters=True, broadcast_buffers=True,
                 process_group=None):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.broadcast_buffers = broadcast_buffers
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world_size > 1:
            self._sync_module_states()
        module._post_backward_hooks.append(self._reduce_gradients)

    def _float_buffers(self):
        return [b for b in self.module.buffers() if b.dtype.is_floating_point]

    def _sync_module_states(self):
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=0, group=self.process_group)

    def _broadcast_buffers(self):
        bufs = self._float_buffers()
        if not bufs:
            return
        flat = torch.cat([b.reshape(-1) for b in bufs])      # memory movement only (coalesced broadcast)
        dist.broadcast(flat, src=0, group=self.process_group)
        off = 0
        with torch.no_grad():
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n

    def _reduce_gradients(self, arena):
        if self.world_size <= 1:
            return
        dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM, group=self.process_group)
        if arena.flat.is_cuda:
            K.axpby(arena.flat, None, 1.0 / self.world_size, 0.0, out=arena.flat)
        else:                                                 # gloo/CPU plumbing tests only
            arena.flat.mul_(1.0 / self.world_size)

    def forward(self, *args, **kwargs):
        if self.world_size > 1 and self.broadcast_buffers and self.module.training and torch.is_grad_enabled():
            self._broadcast_buffers()
        return self.module(*args, **kwargs)
