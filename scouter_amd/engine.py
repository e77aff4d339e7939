"""Epoch loops of the xSlot training path: `train_one_epoch`, `evaluate`, `calculation` (names / arguments /
`record` contents as in the reference's engine.py:6-52).

What is different by design: the reference reads four scalars back to the host every step (loss.item() x3 and the
top-1 accuracy, engine.py:37-42), i.e. four device synchronisations per step.  Here the loss-head kernel already
leaves [loss, nll, area**power, top-1] in one small device buffer (`SlotModel.last_stats`); each step adds that
buffer into a device-side accumulator with one tiny kernel and the host reads the accumulator ONCE per epoch, so the
host never waits for the GPU inside the loop."""
import torch

from . import kernels as K

try:
    from tqdm.auto import tqdm as _progress
except Exception:                                     # pragma: no cover - tqdm is optional
    def _progress(it):
        return it


class _DeviceMeter:
    """Sum of the per-step stats vectors, kept on the device until `.read()`."""

    def __init__(self):
        self.total = None
        self.host = [0.0, 0.0, 0.0, 0.0]              # fallback path for models without `last_stats`

    def add_device(self, stats):
        if self.total is None:
            self.total = torch.zeros(8, dtype=torch.float32, device=stats.device)
        K.axpby(self.total, stats, 1.0, 1.0, out=self.total)

    def add_host(self, loss, nll, att, acc):
        for i, v in enumerate((loss, nll, att, acc)):
            self.host[i] += v

    def read(self):
        dev = self.total[:6].tolist() if self.total is not None else [0.0] * 6      # the only host sync of the epoch
        if dev[5] > 0:                 # stats[5]: labels outside [0, num_classes) seen by the loss kernel this epoch
            raise IndexError("%d target label(s) out of bounds for num_classes (F.nll_loss raises the same way; check "
                             "--num_classes against the dataset)" % int(dev[5]))
        return [d + h for d, h in zip(dev[:4], self.host)]


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


_feed_streams = {}


def device_batches(data_loader, device):
    """Yields (images float32 on `device`, labels int64 on `device`) for every batch of `data_loader`, one batch AHEAD:
    after batch n has been handed out -- i.e. after the caller has queued step n's kernels -- batch n + 1 is taken from
    the loader and its host-to-device copy (ONE pinned buffer, dataset.transform_func.PackedImages) and GPU transform are
    issued on a separate feed stream, where they run under step n; the compute stream then only waits for an event.  The
    counterpart of the reference's DataLoaderX / prefetch_generator thread (reference train.py:158-160,
    tools/prepare_things.py), which prefetches on the host only.  CPU devices (plumbing tests): plain iteration."""
    from .dataset.transform_func import PackedImages
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    feed = None
    if on_gpu:
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        feed = _feed_streams.get(key)
        if feed is None:
            feed = _feed_streams[key] = torch.cuda.Stream(device=device)

    def stage(batch):
        images = batch["image"]
        raw = isinstance(images, (list, tuple, PackedImages))
        if not on_gpu:
            images = data_loader.gpu_transform(images, device) if raw else images.to(device, dtype=torch.float32)
            return images, batch["label"].to(device, dtype=torch.int64), None
        # (no feed.wait_stream(compute): that would order the copy BEHIND step n -- the feed stream has its own allocator
        # pool and workspace, and everything it hands over is fenced by the event + record_stream below)
        with torch.cuda.stream(feed):
            if raw:                                   # decoded uint8 frames: Resize + ToTensor + Normalize on the GPU
                images = data_loader.gpu_transform(images, device)
            else:
                images = images.to(device, dtype=torch.float32, non_blocking=True)
            labels = batch["label"].to(device, dtype=torch.int64, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(feed)
        return images, labels, ev

    it = iter(data_loader)
    try:
        cur = stage(next(it))
    except StopIteration:
        return
    while cur is not None:
        images, labels, ev = cur
        if ev is not None:
            st = torch.cuda.current_stream(device)
            st.wait_event(ev)
            images.record_stream(st)
            labels.record_stream(st)
        yield images, labels
        # (the caller's loop body has run: step n is queued; now fetch and stage batch n + 1 under it)
        try:
            cur = stage(next(it))
        except StopIteration:
            cur = None


def calculation(model, mode, data_loader, device, record, epoch, optimizer=None):
    training = mode == "train"
    meter = _DeviceMeter()
    print("start " + mode + " :" + str(epoch))
    steps = 0
    for images, labels in _progress(device_batches(data_loader, device)):
        if training:
            optimizer.zero_grad()
        logits, losses = model(images, labels)
        if training:
            seed = getattr(_unwrap(model), "loss_seed", None)     # (cached ones: no `ones_like` launch per step)
            losses[0].backward(seed(losses[0])) if seed is not None else losses[0].backward()
            optimizer.step()
        stats = getattr(_unwrap(model), "last_stats", None)
        if stats is not None:
            meter.add_device(stats)
        else:                                         # generic nn.Module: fall back to per-step host reads
            from .tools.calculate_tool import evaluateTop1
            meter.add_host(losses[0].item(), losses[1].item() if len(losses) > 2 else 0.0,
                           losses[2].item() if len(losses) > 2 else 0.0, evaluateTop1(logits, labels))
        steps += 1
    loss, nll, att, acc = (v / max(steps, 1) for v in meter.read())
    for field, value in (("loss", loss), ("acc", acc), ("log_loss", nll), ("att_loss", att)):
        record[mode][field].append(round(value, 3))


def train_one_epoch(model, data_loader, optimizer, device, record, epoch):
    model.train()
    calculation(model, "train", data_loader, device, record, epoch, optimizer)


@torch.no_grad()
def evaluate(model, data_loader, device, record, epoch):
    model.eval()
    calculation(model, "val", data_loader, device, record, epoch)
