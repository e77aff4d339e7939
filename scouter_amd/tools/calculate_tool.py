"""Metrics helpers -- mirrors tools/calculate_tool.py:4-31 of the reference."""
import torch


def evaluateTop1(logits, labels):
    with torch.no_grad():
        pred = logits.argmax(dim=1)
        return torch.eq(pred, labels).sum().float().item() / labels.size(0)


def evaluateTop5(logits, labels):
    with torch.no_grad():
        _, pred = logits.topk(5, 1, True, True)
        return torch.eq(pred, labels.view(-1, 1)).sum().float().item() / labels.size(0)


class MetricLog():
    def __init__(self):
        self.record = {"train": {"loss": [], "acc": [], "log_loss": [], "att_loss": []},
                       "val": {"loss": [], "acc": [], "log_loss": [], "att_loss": []}}

    def print_metric(self):
        names = [("loss", "loss"), ("acc", "acc"), ("log_loss", "CE loss"), ("att_loss", "attention loss")]
        for key, label in names:
            print("train %s:" % label, self.record["train"][key])
            print("val %s:" % label, self.record["val"][key])
