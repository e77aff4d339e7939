"""Training / evaluation loops -- mirrors engine.py:6-52 of the reference (same signatures and `record` contents).

Difference on purpose: the reference pulls four scalars to the host per step (`loss.item()` x3 + evaluateTop1,
engine.py:37-42 = 4 device syncs).  Here the loss head kernel already produced [loss, nll, area, top-1] in one small
device buffer (`model.last_stats`), which is read with a single D2H copy per step."""
import torch

try:
    from tqdm.auto import tqdm
except Exception:   # pragma: no cover
    def tqdm(x):
        return x


def train_one_epoch(model, data_loader, optimizer, device, record, epoch):
    model.train()
    calculation(model, "train", data_loader, device, record, epoch, optimizer)


@torch.no_grad()
def evaluate(model, data_loader, device, record, epoch):
    model.eval()
    calculation(model, "val", data_loader, device, record, epoch)


def _stats_of(model):
    m = model.module if hasattr(model, "module") else model
    return getattr(m, "last_stats", None)


def calculation(model, mode, data_loader, device, record, epoch, optimizer=None):
    L = len(data_loader)
    running_loss = running_corrects = running_att_loss = running_log_loss = 0.0
    print("start " + mode + " :" + str(epoch))
    for i_batch, sample_batch in enumerate(tqdm(data_loader)):
        inputs = sample_batch["image"].to(device, dtype=torch.float32)
        labels = sample_batch["label"].to(device, dtype=torch.int64)
        if mode == "train":
            optimizer.zero_grad()
        logits, loss_list = model(inputs, labels)
        loss = loss_list[0]
        if mode == "train":
            loss.backward()
            optimizer.step()
        stats = _stats_of(model)
        if stats is not None and len(loss_list) > 2:
            s = stats[:4].tolist()                        # one D2H sync: loss, nll, area**power, top-1
            running_loss += s[0]
            running_log_loss += s[1]
            running_att_loss += s[2]
            running_corrects += s[3]
        else:
            from .tools import calculate_tool as cal
            running_loss += loss.item()
            if len(loss_list) > 2:
                running_att_loss += loss_list[2].item()
                running_log_loss += loss_list[1].item()
            running_corrects += cal.evaluateTop1(logits, labels)
    record[mode]["loss"].append(round(running_loss / L, 3))
    record[mode]["acc"].append(round(running_corrects / L, 3))
    record[mode]["log_loss"].append(round(running_log_loss / L, 3))
    record[mode]["att_loss"].append(round(running_att_loss / L, 3))
