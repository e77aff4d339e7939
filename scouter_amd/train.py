"""Training driver of the MI355X xSlot path.

Public surface as in the reference's train.py (`get_args_parser`, `main(args) -> [train_acc, val_acc]`,
`param_translation(args)` with the comma-list sweeps, checkpoint file names and contents), so recipes written for it
run unchanged:

    python -m scouter_amd.train --dataset ImageNet --model resnest26d --channel 2048 --num_classes 10 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m scouter_amd.train ...   # 8 GPUs

One process per GPU; gradients are all-reduced over RCCL/xGMI by scouter_amd.parallel; the optimizer is the fused
AdamW kernel.  The `--thop` FLOP-counting branch of the reference (train.py:91-137) is CPU-side tooling with
third-party packages and is refused."""
import argparse
import datetime
import time
from pathlib import Path

import torch
from torch.utils.data import BatchSampler, DistributedSampler, RandomSampler, SequentialSampler

from .dataset.choose_dataset import select_dataset
from .dataset.transform_func import collate_raw, make_gpu_transform
from .engine import evaluate, train_one_epoch
from .optim import FusedAdamW
from .parallel import DistributedDataParallel
from .sloter.slot_model import SlotModel
from .tools import prepare_things as prt
from .tools.calculate_tool import MetricLog


def _flag(text):
    value = text.lower()
    if value in ("yes", "true", "t", "y", "1"):
        return True
    if value in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Unsupported value encountered.")


# (flag, default, type, help) -- defaults and types are the reference's (train.py:28-78); four flags stay strings so
# that a comma list sweeps them (param_translation)
_CLI = [
    ("model", "resnet18", str, "backbone: resnet18 | resnest26d | resnest50d"),
    ("dataset", "MNIST", str, "MNIST selects the 1-channel 3x3 stem"),
    ("channel", 512, int, "channels of the backbone feature map"),
    ("lr", 1e-4, float, None), ("lr_drop", 70, int, "StepLR period (epochs)"), ("batch_size", 64, int, "per process"),
    ("weight_decay", 1e-4, float, "accepted and ignored, like the reference (AdamW default 1e-2 is used)"),
    ("epochs", 10, int, None), ("num_classes", "10", str, None), ("img_size", 260, None, "input resolution"),
    ("pre_trained", True, _flag, "freeze `freeze_layers` stages (weights must be loaded from a checkpoint)"),
    ("use_slot", True, _flag, "xSlot head (false: global-pool + fc baseline)"),
    ("use_pre", False, _flag, "initialise the backbone from the FC-baseline checkpoint"),
    ("aug", False, _flag, None), ("grad", False, _flag, None), ("grad_min_level", 0.0, float, None),
    ("iterated_evaluation_num", 1, int, "repetitions per sweep value"),
    ("cal_area_size", False, _flag, "tag checkpoints with lambda / slots_per_class"),
    ("thop", False, _flag, "refused (CPU FLOP counting)"),
    ("loss_status", 1, int, "+1 positive / -1 negative explanation loss"),
    ("freeze_layers", 2, int, None), ("hidden_dim", 64, int, None), ("slots_per_class", "3", str, None),
    ("power", "2", str, "exponent of the attention-area loss"), ("to_k_layer", 1, int, None),
    ("lambda_value", "1.", str, "weight of the attention-area loss"),
    ("vis", False, _flag, "write slot attention maps"), ("vis_id", 0, int, None),
    ("dataset_dir", "../PAN/bird_200/CUB_200_2011/CUB_200_2011/", str, None),
    ("output_dir", "saved_model/", str, "empty string: do not save"), ("pre_dir", "pre_model/", str, None),
    ("device", "cuda", str, None), ("num_workers", 4, int, None), ("start_epoch", 0, int, None),
    ("resume", "", str, "checkpoint path"),
    ("world_size", 1, int, None), ("local_rank", None, int, None), ("dist_url", "env://", str, None),
    # additions of this build
    ("precision", "fp32", str, "fp32 | bf16 (backbone convolutions on bf16 matrix inputs, fp32 accumulate; wide bottleneck tensors stored as bf16)"),
    ("synthetic_data", True, _flag, "seeded synthetic batches (the benchmark input)"),
    ("synthetic_len", 256, int, "images per synthetic epoch"),
]


def get_args_parser():
    parser = argparse.ArgumentParser("Set SCOUTER model", add_help=False)
    for name, default, typ, text in _CLI:
        kwargs = {"default": default, "help": text}
        if typ is not None:
            kwargs["type"] = typ
        parser.add_argument("--" + name, **kwargs)
    return parser


def checkpoint_name(args, suffix="checkpoint.pth"):
    """`{dataset}_{use_slot_|no_slot_}[negative_][for_area_size_{lambda}_{spc}_]{suffix}` (reference train.py:180-189)."""
    parts = [args.dataset, "use_slot" if args.use_slot else "no_slot"]
    if args.use_slot and args.loss_status != 1:
        parts.append("negative")
    if args.cal_area_size:
        parts += ["for_area_size", str(args.lambda_value), str(args.slots_per_class)]
    return "_".join(parts) + "_" + suffix


def _build_loaders(args):
    train_set, val_set = select_dataset(args)
    if args.distributed:
        train_sampler, val_sampler = DistributedSampler(train_set), DistributedSampler(val_set, shuffle=False)
    else:
        train_sampler, val_sampler = RandomSampler(train_set), SequentialSampler(val_set)
    raw = not getattr(args, "synthetic_data", True)     # real data: samples are decoded uint8 images of any size
    # raw frames travel packed in one buffer per batch, pinned by the DataLoader's pin thread (one async H2D per batch)
    extra = {"collate_fn": collate_raw, "pin_memory": torch.cuda.is_available()} if raw else {}
    train_loader = prt.DataLoaderX(train_set, batch_sampler=BatchSampler(train_sampler, args.batch_size, drop_last=True),
                                   num_workers=args.num_workers, **extra)
    val_loader = prt.DataLoaderX(val_set, args.batch_size, sampler=val_sampler, num_workers=args.num_workers, **extra)
    if raw:
        train_loader.gpu_transform = val_loader.gpu_transform = make_gpu_transform(args)
    return train_loader, val_loader, train_sampler


def _restore(args, model, optimizer, scheduler):
    state = torch.load(args.resume, map_location="cpu", weights_only=False)
    model.load_state_dict(state["model"])
    if all(k in state for k in ("optimizer", "lr_scheduler", "epoch")):
        try:
            optimizer.load_state_dict(state["optimizer"])
        except Exception as err:      # a per-tensor torch.optim.AdamW state (reference checkpoint) has another layout
            print("optimizer state not restored:", err)
        scheduler.load_state_dict(state["lr_scheduler"])
        args.start_epoch = state["epoch"] + 1


def _save(args, model, optimizer, scheduler, epoch):
    out = Path(args.output_dir)
    names = [checkpoint_name(args)]
    if (epoch + 1) % args.lr_drop == 0 or (epoch + 1) % 10 == 0:      # numbered copy at LR drops / every 10 epochs
        names.append(checkpoint_name(args, "checkpoint%04d.pth" % epoch))
    payload = {"model": model.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": scheduler.state_dict(),
               "epoch": epoch, "args": args}
    for name in names:
        prt.save_on_master(payload, out / name)


def main(args):
    prt.init_distributed_mode(args)
    if args.thop:
        raise NotImplementedError("--thop (CPU FLOP counting with thop / tensorly) is outside the xSlot hot path")
    device = torch.device(args.device)
    net = SlotModel(args).to(device)
    kind = ("use slot " if args.use_slot else "without slot ") + \
        ("negetive loss" if args.use_slot and args.loss_status != 1 else "positive loss")
    print("train model: " + kind)
    wrapped = DistributedDataParallel(net, device_ids=[args.gpu], find_unused_parameters=True) \
        if args.distributed else net
    trainable = [p for p in net.parameters() if p.requires_grad]
    print("number of params:", sum(p.numel() for p in trainable))
    optimizer = FusedAdamW(trainable, lr=args.lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=args.lr_drop)
    train_loader, val_loader, train_sampler = _build_loaders(args)
    if args.resume:
        _restore(args, net, optimizer, scheduler)

    print("Start training")
    started = time.time()
    log = MetricLog()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        train_one_epoch(wrapped, train_loader, optimizer, device, log.record, epoch)
        scheduler.step()
        if args.output_dir:
            _save(args, net, optimizer, scheduler, epoch)
        evaluate(wrapped, val_loader, device, log.record, epoch)
        log.print_metric()
    print("Training time {}".format(datetime.timedelta(seconds=int(time.time() - started))))
    return [log.record["train"]["acc"][-1], log.record["val"]["acc"][-1]]


_SWEEPABLE = (("num_classes", int), ("lambda_value", float), ("power", int), ("slots_per_class", int))


def param_translation(args):
    """Casts the four string flags; a comma list in ONE of them runs main() once per value (x iterated_evaluation_num)
    and returns {flag-value: [main() results]} (reference train.py:207-231)."""
    sweep = None
    for name, cast in _SWEEPABLE:
        raw = str(getattr(args, name))
        if "," in raw:
            sweep = (name, cast, raw.split(","))
        else:
            setattr(args, name, cast(raw))
    if sweep is None:
        return main(args)
    name, cast, values = sweep
    results = {}
    for value in values:
        setattr(args, name, cast(value))
        runs = results.setdefault("%s-%s" % (name, value), [])
        for _ in range(args.iterated_evaluation_num):
            runs.append(main(args))
            print(results)
    return results


if __name__ == "__main__":
    cli = argparse.ArgumentParser("model training and evaluation script", parents=[get_args_parser()]).parse_args()
    if cli.output_dir:
        Path(cli.output_dir).mkdir(parents=True, exist_ok=True)
    param_translation(cli)
