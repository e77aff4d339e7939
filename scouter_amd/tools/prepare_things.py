"""Distributed bootstrap helpers -- mirrors tools/prepare_things.py:9-75 of the reference (same env-var contract:
RANK / WORLD_SIZE / LOCAL_RANK, or SLURM_PROCID).  backend 'nccl' is RCCL on ROCm (collectives ride xGMI)."""
import os

import torch
import torch.distributed as dist


def init_distributed_mode(args):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ["LOCAL_RANK"])
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % torch.cuda.device_count()
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available() and str(getattr(args, "device", "cuda")).startswith("cuda")
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    print("| distributed init (rank {}): {}".format(args.rank, args.dist_url), flush=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def setup_for_distributed(is_master):
    """Disables printing on non-master ranks (prepare_things.py:34-46)."""
    import builtins as __builtin__
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


class DataLoaderX(torch.utils.data.DataLoader):
    """The reference wraps the iterator in prefetch_generator.BackgroundGenerator (prepare_things.py:140-142);
    that package is not a dependency here -- DataLoader's own worker prefetch is used."""
