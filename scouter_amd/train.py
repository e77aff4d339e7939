"""Training driver -- mirrors train.py:18-238 of the reference: same argparse surface (get_args_parser), same
main(args) flow (init dist -> SlotModel -> DP wrap -> AdamW/StepLR -> loaders -> epoch loop -> checkpoints) and
checkpoint naming.  Launch: `python -m scouter_amd.train ...` or, for N GPUs of one node,
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m scouter_amd.train ...` (one process
per GPU, gradients all-reduced over RCCL/xGMI).  The `--thop` cost-counting branch (train.py:91-137) is out of
scope (it profiles FLOPs on CPU with third-party packages)."""
import argparse
import datetime
import time
from pathlib import Path

import torch
from torch.utils.data import DistributedSampler

from .dataset.choose_dataset import select_dataset
from .engine import evaluate, train_one_epoch
from .optim import FusedAdamW
from .parallel import DistributedDataParallel
from .sloter.slot_model import SlotModel
from .tools import prepare_things as prt
from .tools.calculate_tool import MetricLog
from .tools.prepare_things import DataLoaderX


def get_args_parser():
    def str2bool(v):
        if v.lower() in ("yes", "true", "t", "y", "1"):
            return True
        if v.lower() in ("no", "false", "f", "n", "0"):
            return False
        raise argparse.ArgumentTypeError("Unsupported value encountered.")

    p = argparse.ArgumentParser("Set SCOUTER model", add_help=False)
    p.add_argument("--model", default="resnet18", type=str)
    p.add_argument("--dataset", default="MNIST", type=str)
    p.add_argument("--channel", default=512, type=int)
    # training set
    p.add_argument("--lr", default=0.0001, type=float)
    p.add_argument("--lr_drop", default=70, type=int)
    p.add_argument("--batch_size", default=64, type=int)
    p.add_argument("--weight_decay", default=0.0001, type=float)       # unused by the reference too (train.py:146)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--num_classes", default="10", type=str)
    p.add_argument("--img_size", default=260, help="input resolution")
    p.add_argument("--pre_trained", default=True, type=str2bool)
    p.add_argument("--use_slot", default=True, type=str2bool)
    p.add_argument("--use_pre", default=False, type=str2bool)
    p.add_argument("--aug", default=False, type=str2bool)
    p.add_argument("--grad", default=False, type=str2bool)
    p.add_argument("--grad_min_level", default=0., type=float)
    p.add_argument("--iterated_evaluation_num", default=1, type=int)
    p.add_argument("--cal_area_size", default=False, type=str2bool)
    p.add_argument("--thop", default=False, type=str2bool)
    # slot setting
    p.add_argument("--loss_status", default=1, type=int)
    p.add_argument("--freeze_layers", default=2, type=int)
    p.add_argument("--hidden_dim", default=64, type=int)
    p.add_argument("--slots_per_class", default="3", type=str)
    p.add_argument("--power", default="2", type=str)
    p.add_argument("--to_k_layer", default=1, type=int)
    p.add_argument("--lambda_value", default="1.", type=str)
    p.add_argument("--vis", default=False, type=str2bool)
    p.add_argument("--vis_id", default=0, type=int)
    # data / machine set
    p.add_argument("--dataset_dir", default="../PAN/bird_200/CUB_200_2011/CUB_200_2011/")
    p.add_argument("--output_dir", default="saved_model/")
    p.add_argument("--pre_dir", default="pre_model/")
    p.add_argument("--device", default="cuda")
    p.add_argument("--num_workers", default=4, type=int)
    p.add_argument("--start_epoch", default=0, type=int, metavar="N")
    p.add_argument("--resume", default="", type=str, help="checkpoint path to resume from")
    # distributed training parameters
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", type=int)
    p.add_argument("--dist_url", default="env://")
    # build-side additions (not in the reference)
    p.add_argument("--synthetic_data", default=True, type=str2bool, help="seeded synthetic batches (benchmark input)")
    p.add_argument("--synthetic_len", default=256, type=int)
    return p


def checkpoint_name(args, suffix="checkpoint.pth"):
    """reference train.py:180-189"""
    return (f"{args.dataset}_" + ("use_slot_" if args.use_slot else "no_slot_")
            + ("negative_" if args.use_slot and args.loss_status != 1 else "")
            + (f"for_area_size_{args.lambda_value}_{args.slots_per_class}_" if args.cal_area_size else "") + suffix)


def main(args):
    prt.init_distributed_mode(args)
    device = torch.device(args.device)
    if args.thop:
        raise NotImplementedError("--thop (CPU FLOP counting with thop/tensorly) is outside the xSlot hot path")
    model = SlotModel(args)
    print("train model: " + ("use slot " if args.use_slot else "without slot ")
          + ("negetive loss" if args.use_slot and args.loss_status != 1 else "positive loss"))
    model.to(device)
    model_without_ddp = model
    if args.distributed:
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=True)
        model_without_ddp = model.module
    print("number of params:", sum(p.numel() for p in model.parameters() if p.requires_grad))
    params = [p for p in model_without_ddp.parameters() if p.requires_grad]
    optimizer = FusedAdamW(params, lr=args.lr)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=args.lr_drop)

    dataset_train, dataset_val = select_dataset(args)
    if args.distributed:
        sampler_train = DistributedSampler(dataset_train)
        sampler_val = DistributedSampler(dataset_val, shuffle=False)
    else:
        sampler_train = torch.utils.data.RandomSampler(dataset_train)
        sampler_val = torch.utils.data.SequentialSampler(dataset_val)
    batch_sampler_train = torch.utils.data.BatchSampler(sampler_train, args.batch_size, drop_last=True)
    data_loader_train = DataLoaderX(dataset_train, batch_sampler=batch_sampler_train, num_workers=args.num_workers)
    data_loader_val = DataLoaderX(dataset_val, args.batch_size, sampler=sampler_val, num_workers=args.num_workers)
    output_dir = Path(args.output_dir) if args.output_dir else None

    if args.resume:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        model_without_ddp.load_state_dict(checkpoint["model"])
        if "optimizer" in checkpoint and "lr_scheduler" in checkpoint and "epoch" in checkpoint:
            try:
                optimizer.load_state_dict(checkpoint["optimizer"])
            except Exception as e:          # a reference (per-tensor AdamW) optimizer state has a different layout
                print("optimizer state not restored:", e)
            lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
            args.start_epoch = checkpoint["epoch"] + 1

    print("Start training")
    start_time = time.time()
    log = MetricLog()
    record = log.record
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            sampler_train.set_epoch(epoch)
        train_one_epoch(model, data_loader_train, optimizer, device, record, epoch)
        lr_scheduler.step()
        if output_dir is not None:
            paths = [output_dir / checkpoint_name(args)]
            if (epoch + 1) % args.lr_drop == 0 or (epoch + 1) % 10 == 0:
                paths.append(output_dir / checkpoint_name(args, f"checkpoint{epoch:04}.pth"))
            for path in paths:
                prt.save_on_master({"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(),
                                    "lr_scheduler": lr_scheduler.state_dict(), "epoch": epoch, "args": args}, path)
        evaluate(model, data_loader_val, device, record, epoch)
        log.print_metric()
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    return [record["train"]["acc"][-1], record["val"]["acc"][-1]]


def param_translation(args):
    """reference train.py:207-231: four flags are strings so that a comma list sweeps one of them."""
    args_dict = vars(args)
    names, types = ["num_classes", "lambda_value", "power", "slots_per_class"], [int, float, int, int]
    target, target_type, settings = None, None, None
    for name, typ in zip(names, types):
        if str(args_dict[name]).find(",") > 0:
            target, target_type, settings = name, typ, str(args_dict[name]).split(",")
        else:
            args_dict[name] = typ(args_dict[name])
    if target is None:
        return main(args)
    record = {}
    for s in settings:
        record[f"{target}-" + s] = []
        args_dict[target] = target_type(s)
        for _ in range(args.iterated_evaluation_num):
            record[f"{target}-" + s].append(main(args))
            print(record)
    return record


if __name__ == "__main__":
    parser = argparse.ArgumentParser("model training and evaluation script", parents=[get_args_parser()])
    args = parser.parse_args()
    if args.output_dir:
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    param_translation(args)
