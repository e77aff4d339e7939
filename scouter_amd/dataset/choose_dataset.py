"""Dataset selection (reference dataset/choose_dataset.py:7-28).  `--synthetic_data true` (default): seeded synthetic
batches {"image": float [B,c,H,W], "label": int64 [B]} -- the benchmark input of BASELINE.md section 3.
`--synthetic_data false`: the reference's readers (MNIST, CUB200, ConText, ImageNet folders); their samples carry the
DECODED uint8 image, and the loader applies Resize + ToTensor + Normalize on the GPU (dataset/transform_func.py)."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticImages(Dataset):
    def __init__(self, n, channels, size, num_classes, seed):
        self.n, self.c, self.size, self.num_classes, self.seed = n, channels, size, num_classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed + i)
        img = rng.standard_normal((self.c, self.size, self.size), dtype=np.float32)
        return {"image": torch.from_numpy(img), "label": int(rng.integers(0, self.num_classes)), "names": str(i)}


def select_dataset(args):
    if getattr(args, "dataset", "") not in ("synthetic", "MNIST", "ImageNet", "CUB200", "ConText"):
        raise ValueError("unknown dataset %s" % args.dataset)
    if args.dataset != "synthetic" and not getattr(args, "synthetic_data", True):
        return select_real_dataset(args)
    c = 1 if args.dataset == "MNIST" else 3
    size = int(args.img_size)
    n_train = int(getattr(args, "synthetic_len", 4 * args.batch_size))
    n_cls = int(args.num_classes)
    return SyntheticImages(n_train, c, size, n_cls, 1234), SyntheticImages(max(args.batch_size, n_train // 4), c, size,
                                                                           n_cls, 987654)


def select_real_dataset(args):
    from .transform_func import make_transform
    if args.dataset == "MNIST":
        from .mnist import MNIST
        root = getattr(args, "dataset_dir", None) or "./data/mnist"
        if not os.path.isdir(os.path.join(root, "MNIST")):
            root = "./data/mnist"                               # the reference hard-codes this location
        return (MNIST(root, train=True, transform=make_transform(args, "train")),
                MNIST(root, train=False, transform=make_transform(args, "val")))
    if args.dataset == "CUB200":
        from .CUB200 import CUB_200
        return (CUB_200(args, train=True, transform=make_transform(args, "train")),
                CUB_200(args, train=False, transform=make_transform(args, "val")))
    from .ConText import ConText, MakeList, MakeListImage
    if args.dataset == "ConText":
        train, val = MakeList(args).get_data()
    elif args.dataset == "ImageNet":
        train, val = MakeListImage(args).get_data()
    else:
        raise ValueError(f"unknown {args.dataset}")
    return ConText(train, transform=make_transform(args, "train")), ConText(val, transform=make_transform(args, "val"))
