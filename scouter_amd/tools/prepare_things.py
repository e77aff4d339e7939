"""Process-group bootstrap and rank helpers for one-process-per-GPU data parallelism.

Same contract as the reference's tools/prepare_things.py:9-75: `init_distributed_mode(args)` reads
RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run) or SLURM_PROCID, fills args.rank / world_size / gpu /
distributed / dist_backend, initialises the process group and silences print() on non-master ranks.  On ROCm the
'nccl' backend is RCCL (collectives over xGMI); without a GPU ('--device cpu', tests) it falls back to gloo."""
import builtins
import os

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader


def _rank_from_env():
    """(rank, world_size, local_rank) or None when the process was not launched by a distributed launcher."""
    env = os.environ
    if "RANK" in env and "WORLD_SIZE" in env:
        return int(env["RANK"]), int(env["WORLD_SIZE"]), int(env.get("LOCAL_RANK", 0))
    if "SLURM_PROCID" in env:
        rank = int(env["SLURM_PROCID"])
        return rank, int(env.get("SLURM_NTASKS", getattr(torch.cuda, "device_count", lambda: 1)() or 1)), \
            rank % max(torch.cuda.device_count(), 1)
    return None


def setup_for_distributed(is_master):
    """print() becomes a no-op on non-master ranks unless called with force=True."""
    plain_print = builtins.print

    def rank_aware_print(*args, force=False, **kwargs):
        if is_master or force:
            plain_print(*args, **kwargs)

    builtins.print = rank_aware_print


def init_distributed_mode(args):
    found = _rank_from_env()
    if found is None:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.rank, env_world, args.gpu = found
    args.world_size = env_world
    args.distributed = True
    on_gpu = torch.cuda.is_available() and str(getattr(args, "device", "cuda")).startswith("cuda")
    if on_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if on_gpu else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    print("| distributed init (rank {}): {}".format(args.rank, args.dist_url), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(obj, path, **kwargs):
    if is_main_process():
        torch.save(obj, path, **kwargs)


def get_name(root, mode_folder=True):
    """Sorted sub-directory names (mode_folder) or file names of `root` itself (reference prepare_things.py:145-150)."""
    for _, dirs, files in os.walk(root):
        return sorted(dirs) if mode_folder else sorted(files)
    return None


class DataLoaderX(DataLoader):
    """DataLoader whose batches may carry raw decoded images: `gpu_transform` (dataset.transform_func.GpuTransform),
    when set, is applied by the engine on the device.  (The reference adds a prefetch_generator background thread;
    the DataLoader's own worker prefetching is used here.)"""
    gpu_transform = None
