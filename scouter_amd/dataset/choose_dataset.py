"""Dataset selection.  The reference's readers (dataset/*.py: CUB200, ConText, MNIST, ImageNet folders + imgaug) are
host-side data plumbing outside the xSlot hot path; the hot path only depends on the
batch contract {"image": float [B,c,H,W], "label": int64 [B]} (dataset/mnist.py:102, CUB200.py:76).  This module
provides that contract over seeded synthetic data (the benchmark input of BASELINE.md section 3)."""
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
        raise NotImplementedError("real-data readers are outside the xSlot hot path (SURVEY.md section 8f item 4); "
                                  "pass --synthetic_data true")
    c = 1 if args.dataset == "MNIST" else 3
    size = int(args.img_size)
    n_train = int(getattr(args, "synthetic_len", 4 * args.batch_size))
    n_cls = int(args.num_classes)
    return SyntheticImages(n_train, c, size, n_cls, 1234), SyntheticImages(max(args.batch_size, n_train // 4), c, size,
                                                                           n_cls, 987654)
