"""Classification metrics and the per-epoch record used by the engine.

Public names follow the reference (`tools/calculate_tool.py`: evaluateTop1, evaluateTop5, MetricLog) so that code
written against it keeps working; the implementation is a single top-k routine plus a small record class."""
import torch

_SPLITS = ("train", "val")
_FIELDS = ("loss", "acc", "log_loss", "att_loss")
_TITLES = {"loss": "loss", "acc": "acc", "log_loss": "CE loss", "att_loss": "attention loss"}


@torch.no_grad()
def topk_accuracy(logits, labels, k=1):
    """Fraction of rows whose label is among the k largest logits (python float; one host sync)."""
    k = min(int(k), logits.shape[1])
    hits = logits.topk(k, dim=1).indices.eq(labels.reshape(-1, 1)).any(dim=1)
    return hits.float().mean().item()


def evaluateTop1(logits, labels):
    return topk_accuracy(logits, labels, 1)


def evaluateTop5(logits, labels):
    return topk_accuracy(logits, labels, 5)


class MetricLog:
    """record[split][field] -> list with one (3-decimal rounded) entry per epoch."""

    def __init__(self):
        self.record = {split: {field: [] for field in _FIELDS} for split in _SPLITS}

    def append(self, split, loss, acc, log_loss, att_loss):
        for field, value in zip(_FIELDS, (loss, acc, log_loss, att_loss)):
            self.record[split][field].append(round(float(value), 3))

    def print_metric(self):
        for field in _FIELDS:
            for split in _SPLITS:
                print("%s %s:" % (split, _TITLES[field]), self.record[split][field])
