"""Data-parallel training over RCCL/xGMI -- the MI355X counterpart of the reference's
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)` (train.py:139-141).

One process per GPU (torch.distributed, backend 'nccl' == RCCL on ROCm).  Because every parameter gradient of the HIP
path lives in ONE flat fp32 arena, the per-step exchange is a single sum-all-reduce of that buffer (43-98 MiB)
followed by a scale by 1/world -- no bucketing machinery, no graph walk for unused parameters (`to_q` is excluded
statically).  Like DDP: parameters and buffers are broadcast from rank 0 at construction, and BatchNorm running
statistics are re-broadcast from rank 0 in front of EVERY forward (`broadcast_buffers=True`; asynchronously, consumed in
front of the first BatchNorm)."""
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
        self._active = dist.is_initialized()          # a 1-rank group still runs the collectives (used by tests)
        if self._active:
            self._sync_module_states()
        self._pending = []
        self._buckets = 0
        self.buckets_last_step = 0
        self._buf_flat = None
        self._buf_work = None            # this forward's asynchronous buffer broadcast (consumed in front of the first BatchNorm)
        self.measure_exposed = False     # bench.py: hipEvent pairs around the wait for the gradient collectives
        self.exposed_events = []
        if self._active and broadcast_buffers:
            self._flatten_float_buffers()
        module._grad_ready_hooks.append(self._launch_bucket)
        module._post_backward_hooks.append(self._finish_gradients)
        # the backbone calls these right before its first BatchNorm touches the running statistics
        bone = getattr(module, "backbone", None)
        self._bn_hooked = bone is not None and hasattr(bone, "_pre_bn_hooks")
        if self._bn_hooked:
            bone._pre_bn_hooks.append(self._consume_buffer_broadcast)

    def _float_buffers(self):
        return [b for b in self.module.buffers() if b.dtype.is_floating_point]

    @staticmethod
    def _dense(t):
        """A contiguous tensor sharing t's storage (conv weights are logical-OIHW views over an HWIO buffer)."""
        if t.is_contiguous():
            return t
        if t.dim() == 4 and t.permute(2, 3, 1, 0).is_contiguous():
            return t.permute(2, 3, 1, 0)
        raise RuntimeError("parameter with an unexpected memory layout: %s / %s" % (tuple(t.shape), t.stride()))

    def _coalesced_broadcast(self, tensors):
        """One broadcast from rank 0 for a list of same-dtype tensors (memory movement only)."""
        if not tensors:
            return
        dense = [self._dense(t.data) for t in tensors]
        flat = torch.cat([d.reshape(-1) for d in dense])
        dist.broadcast(flat, src=0, group=self.process_group)
        off = 0
        with torch.no_grad():
            for d in dense:
                n = d.numel()
                d.copy_(flat[off:off + n].view_as(d))
                off += n

    def _sync_module_states(self):
        ts = list(self.module.parameters()) + list(self.module.buffers())
        for dtype in sorted({t.dtype for t in ts}, key=str):
            self._coalesced_broadcast([t for t in ts if t.dtype == dtype])

    def _flatten_float_buffers(self):
        """Re-point every floating-point buffer (BatchNorm running statistics) at a slice of ONE flat tensor so the
        per-step DDP-style buffer broadcast is a single collective without gather/scatter copies."""
        named = [(m, n, b) for m in self.module.modules() for n, b in m._buffers.items()
                 if b is not None and b.dtype == torch.float32]
        if not named:
            return
        flat = torch.cat([b.detach().reshape(-1) for _, _, b in named])
        off = 0
        for m, n, b in named:
            k = b.numel()
            m._buffers[n] = flat[off:off + k].view(b.shape)
            off += k
        self._buf_flat, self._buf_first = flat, named[0][0]._buffers[named[0][1]]

    def _buffers_are_flat(self):
        return self._buf_flat is not None and self._buf_first.data_ptr() == self._buf_flat.data_ptr()

    def _issue_buffer_broadcast(self):
        """DDP's per-forward buffer broadcast from rank 0 (BatchNorm running statistics; torch DDP does it in front of
        EVERY forward, training or not).  The same ONE collective on every rank in every forward -- no rank-local decision
        (round 4 skipped it when the tensor version counter said nothing had written the buffers: a rank-0-only
        load_state_dict then left rank 0 issuing a collective the other ranks did not) -- but issued
        ASYNCHRONOUSLY on RCCL's stream: the first kernels of the forward (stem im2col + GEMM) do not touch the running
        statistics, so the compute stream only waits for it in front of the first BatchNorm (`_consume_buffer_broadcast`,
        called by the backbone through `_pre_bn_hooks`) and the collective's latency runs under the stem convolution.
        Nothing is in flight between two forwards: state_dict / load_state_dict / eval never race with the collective."""
        self._consume_buffer_broadcast()
        if self._buffers_are_flat():
            self._buf_work = dist.broadcast(self._buf_flat, src=0, group=self.process_group, async_op=True)
        else:                                   # buffers were re-allocated (e.g. module.to(...)): gather / scatter
            self._coalesced_broadcast(self._float_buffers())

    def _consume_buffer_broadcast(self):
        work, self._buf_work = self._buf_work, None
        if work is not None:
            work.wait()                         # compute stream waits for the collective (no host sync on NCCL)

    def _launch_bucket(self, arena, lo, hi):
        """arena.flat[lo:hi] is final: start its sum-all-reduce now (asynchronously on RCCL's stream, ordered after
        the kernels already queued on the compute stream) so that it overlaps the rest of the backward pass."""
        if not self._active or hi <= lo:
            return
        self._pending.append(dist.all_reduce(arena.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.process_group,
                                             async_op=True))
        self._buckets += 1

    def _finish_gradients(self, arena):
        if not self._active:
            return
        ev = None
        if self.measure_exposed and arena.flat.is_cuda:
            # what the compute stream loses to the gradient exchange: from "every backward kernel is queued" to "the last
            # bucket has arrived" on the compute stream (0 when the buckets finished under the backward)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work in self._pending:
            work.wait()                  # makes the compute stream wait for the collective (no host sync on NCCL)
        self._pending = []
        self.buckets_last_step, self._buckets = self._buckets, 0     # (gradient collectives this backward issued: bench.py)
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
        if arena.flat.is_cuda:
            K.axpby(arena.flat, None, 1.0 / self.world_size, 0.0, out=arena.flat)
        else:                                                 # gloo/CPU plumbing tests only
            arena.flat.mul_(1.0 / self.world_size)

    def exposed_allreduce_ms(self):
        """Mean per-step time the compute stream waited for the gradient collectives (measure_exposed; call after a
        device synchronisation) and clears the log."""
        evs, self.exposed_events = self.exposed_events, []
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs) if evs else None

    def forward(self, *args, **kwargs):
        if self._active and self.broadcast_buffers:
            self._issue_buffer_broadcast()
            if not self._bn_hooked:
                # no backbone hook in front of the first BatchNorm (a module that is not a ResNet SlotModel): nothing tells us
                # where the buffers are first touched, so the forward starts behind the collective
                self._consume_buffer_broadcast()
        try:
            return self.module(*args, **kwargs)
        finally:
            self._consume_buffer_broadcast()   # (no-op when the backbone's first BatchNorm already waited for it)
