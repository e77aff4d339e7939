"""hipGraph capture of the steady-state training step.

One step of the HIP path is ~520 kernel launches issued from Python (7-8 ms of host time); on the small configurations
(BASELINE configs[0]: 8.8 ms of GPU time per step) and in bf16 mode the host is the limiter, and even the GPU-bound
headline step loses ~0.6 ms to inter-kernel gaps.  Shapes are static in training (the reference's BatchSampler has
drop_last=True, train.py:155-156) and every buffer of the path has a fixed role, so the whole step

    zero_grad -> forward -> loss -> backward (both streams) -> [gradient all-reduce] -> FusedAdamW

is recorded ONCE into a hipGraph and replayed: one launch call per step.  What makes the step capturable:
  * no host synchronisation inside it (metrics stay on the device, engine.py; block tiles are tuned during warm-up),
  * FusedAdamW(capturable=True): learning rate and step count live in device memory (optim.py),
  * activations / workspaces are allocated from the graph's private memory pool during capture, so replays see the same
    addresses; inputs are copied into static buffers before each replay; outputs are static tensors that the NEXT replay
    overwrites (read them -- or `.clone()` -- before calling again),
  * the weight-gradient side stream forks from and rejoins the capturing stream (kernels.side_stream).
The kernels, their order and their arithmetic are exactly those of the eager step: results are bit-identical
(tests/test_graph_gpu.py)."""
import torch

from . import kernels as K


class GraphedTrainStep:
    """`step = GraphedTrainStep(net, optimizer); log_probs, stats = step(images, labels)`.

    net: SlotModel, or scouter_amd.parallel.DistributedDataParallel around one (then the per-stage gradient all-reduces
    and the BatchNorm-buffer broadcast are part of the graph -- RCCL collectives are capturable; set
    `capture_collectives=False` to keep the data-parallel wrapper eager and use the graph on single-GPU runs only).
    stats = SlotModel.last_stats ([loss, nll, area**power, top-1, area, bad-label count, ...])."""

    def __init__(self, net, optimizer, warmup=3, preserve_state=True):
        """preserve_state: capture needs `warmup` real executions of the step (tile tuning, lazy buffers); with
        preserve_state the parameters, buffers and optimizer state are snapshotted before and restored after, so the first
        replay is training step 1 exactly as in the eager loop."""
        self.preserve_state = bool(preserve_state)
        self.net = net
        self.model = net.module if hasattr(net, "module") else net
        self.optimizer = optimizer
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FusedAdamW(..., capturable=True)")
        self.warmup = max(int(warmup), 2)
        self.graph = None
        self._shape = None
        self.x = self.y = self.logp = self.stats = None
        self.replays = 0

    # the step exactly as engine.calculation runs it, without the autograd engine in between
    def _eager(self, x, y):
        model = self.model
        ddp = hasattr(self.net, "_issue_buffer_broadcast") and self.net._active and self.net.broadcast_buffers
        if ddp:
            self.net._issue_buffer_broadcast()      # consumed by the backbone in front of its first BatchNorm
            if not getattr(self.net, "_bn_hooked", False):
                self.net._consume_buffer_broadcast()    # (no backbone hook: the forward starts behind the collective)
        model.grad_arena()
        logp, stats, state = model._forward_impl(x, y, save=True)
        if ddp:
            self.net._consume_buffer_broadcast()
        model._backward_impl(state, None, self._one, None, None)
        self.optimizer.step()
        return logp, stats

    def _capture(self, images, labels):
        dev = images.device
        self._one = torch.ones(1, dtype=torch.float32, device=dev)
        self.x = torch.empty_like(images, dtype=torch.float32).contiguous()
        self.y = torch.empty_like(labels, dtype=torch.int64)
        self.x.copy_(images)
        self.y.copy_(labels)
        self.model.train()
        snap = self._snapshot() if self.preserve_state else None
        # warm-up on a side stream (as torch.cuda.graphs recommends): tile autotuning, lazy caches, optimizer plan, the
        # device-side hyper-parameter buffers -- everything that must not happen during capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager(self.x, self.y)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        ws_before, arr_before = set(K._ws), set(K._arrival)
        with torch.cuda.graph(self.graph):
            self.logp, self.stats = self._eager(self.x, self.y)
        # Every scratch buffer the captured kernels may address stays alive as long as the graph does: the
        # persistent side streams ("wgrad", "branch") re-use workspaces that existed BEFORE the capture, so the graph has
        # their addresses baked in -- if a later eager call on those streams needed a bigger workspace, kernels.workspace()
        # would replace the buffer and the caching allocator could hand the old one to someone else under the replays.
        self._pinned_ws = list(K._ws.values())
        sp = getattr(self.model, "_wsplitter", None)          # plane-weight tables / plane buffers of the captured key
        self._pinned_ws += sp.buffers() if sp is not None else []
        self._pinned_ws += list(K._arrival.values())          # arrival counters of the split-K weight gradients
        for key in set(K._ws) - ws_before:        # scratch allocated from the graph's private pool: never hand it to
            del K._ws[key]                        # eager code that happens to run on a stream with the same handle
        for key in set(K._arrival) - arr_before:
            del K._arrival[key]
        self._shape = (tuple(images.shape), tuple(labels.shape))
        # the capture pass itself only RECORDED a step; the host-side step mirror advanced though: take that back
        for gi, _ in enumerate(self.optimizer.param_groups):
            st = self.optimizer.state.get("_flat_%d" % gi)
            if st is not None:
                st["step"] -= 1
        if snap is not None:
            self._restore(snap)
        return 0 if snap is not None else self.warmup

    def _named_state(self):
        named = dict(self.model.named_parameters())
        named.update(dict(self.model.named_buffers()))
        return named

    def _snapshot(self):
        opt = {}
        for key, st in self.optimizer.state.items():
            if isinstance(key, str) and key.startswith("_flat_"):
                opt[key] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        return {k: t.detach().clone() for k, t in self._named_state().items()}, opt

    def _restore(self, snap):
        copies, opt = snap
        # by NAME, into the tensors the model holds NOW: the first training forward re-points the BatchNorm step counters
        # at views of one flat buffer (SlotModel._bump_tracked), so the tensor objects of the snapshot may be stale
        live = self._named_state()
        with torch.no_grad():
            for k, c in copies.items():
                live[k].copy_(c)                             # in place: the graph holds these addresses
            for gi, group in enumerate(self.optimizer.param_groups):
                key = "_flat_%d" % gi
                st = self.optimizer.state.get(key)
                if st is None:
                    continue
                old = opt.get(key)
                for name in ("exp_avg", "exp_avg_sq"):
                    if old is not None and name in old:
                        st[name].copy_(old[name])
                    else:
                        st[name].zero_()                      # the state did not exist before the warm-up
                st["step"] = int(old["step"]) if old is not None else 0
                dyn = self.optimizer._dyn.get(gi)
                if dyn is not None:
                    dyn[1:2].fill_(float(st["step"]))

    def __call__(self, images, labels):
        if not images.is_cuda:
            raise RuntimeError("GraphedTrainStep runs on a HIP device (got %s)" % images.device)
        if self.graph is None:
            done = self._capture(images, labels)
            # the warm-up steps already trained on this batch `done` times; one more replay would be step done + 1 --
            # callers that need exactly one update per call use warmup through `prepare()` on a throw-away batch
            self._warm_steps = done
        elif (tuple(images.shape), tuple(labels.shape)) != self._shape:
            raise RuntimeError("GraphedTrainStep was captured for shapes %s, got %s -- static shapes only (use the eager "
                               "step for a ragged last batch)" % (self._shape, (tuple(images.shape), tuple(labels.shape))))
        self.x.copy_(images)
        self.y.copy_(labels)
        self.optimizer.sync_hyperparameters(replays=1)
        self.graph.replay()
        self.replays += 1
        self.model.last_stats = self.stats
        return self.logp, self.stats

    def prepare(self, images, labels):
        """Capture now (running `warmup` real optimisation steps on this batch) without replaying."""
        if self.graph is None:
            self._capture(images, labels)
        return self
