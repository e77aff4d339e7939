"""MNIST reader (reference dataset/mnist.py:14-106 -- a copy of torchvision's).  Sources, in this order:
`<root>/MNIST/processed/{training,test}.pt` (what the reference's `download=True` leaves behind) or the raw IDX files
`<root>/MNIST/raw/{train,t10k}-{images-idx3,labels-idx1}-ubyte[.gz]`.  There is no network here, so nothing is
downloaded: a missing dataset raises, like the reference's `_check_exists` failure.  Samples: {"image", "label"} with
the image handed to `transform` as a mode-'L' PIL image (mnist.py:93-102)."""
import gzip
import os
import struct

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def read_idx(path):
    """IDX ('SN3 Pascal Vincent') file -> numpy array; unsigned-byte payloads only (MNIST images and labels)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, dtype_code, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0 or dtype_code != 8 or not 1 <= ndim <= 3:
        raise ValueError("%s: not an unsigned-byte IDX file" % path)
    dims = struct.unpack(">" + "I" * ndim, raw[4:4 + 4 * ndim])
    return np.frombuffer(raw, dtype=np.uint8, offset=4 + 4 * ndim).reshape(dims)


class MNIST(Dataset):
    training_file, test_file = "training.pt", "test.pt"
    classes = ["%d - %s" % (i, n) for i, n in enumerate(
        ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine"])]

    def __init__(self, root, train=True, transform=None, target_transform=None, download=False):
        self.root, self.train, self.transform, self.target_transform = root, train, transform, target_transform
        processed = os.path.join(self.processed_folder, self.training_file if train else self.test_file)
        if os.path.exists(processed):
            data, targets = torch.load(processed, weights_only=False)
            self.data, self.targets = np.asarray(data, dtype=np.uint8), np.asarray(targets, dtype=np.int64)
        else:
            stem = "train" if train else "t10k"
            imgs = self._find(stem + "-images-idx3-ubyte")
            labs = self._find(stem + "-labels-idx1-ubyte")
            if imgs is None or labs is None:
                raise RuntimeError("Dataset not found under %s (no network: place the MNIST files there)" % self.root)
            self.data, self.targets = read_idx(imgs), read_idx(labs).astype(np.int64)

    def _find(self, name):
        for cand in (os.path.join(self.raw_folder, name), os.path.join(self.raw_folder, name + ".gz")):
            if os.path.exists(cand):
                return cand
        return None

    @property
    def raw_folder(self):
        return os.path.join(self.root, self.__class__.__name__, "raw")

    @property
    def processed_folder(self):
        return os.path.join(self.root, self.__class__.__name__, "processed")

    def __getitem__(self, index):
        img, target = Image.fromarray(self.data[index], mode="L"), int(self.targets[index])
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return {"image": img, "label": target}

    def __len__(self):
        return len(self.data)
