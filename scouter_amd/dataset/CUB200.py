"""CUB-200-2011 reader (reference dataset/CUB200.py:8-82): the split / label / path text files of the dataset root,
classes 1..num_classes only (the 3-digit prefix of the image folder), 0-based labels, grayscale files promoted to RGB.
Samples are {"image", "label", "names"} like the reference's; the image is whatever `transform` returns (here: the
decoded uint8 array -- resize / normalise happen on the GPU, dataset/transform_func.py)."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def _pairs(path):
    with open(path) as f:
        return [line.split() for line in f if line.strip()]


class CUB_200(Dataset):
    def __init__(self, args, train=True, transform=None):
        super().__init__()
        self.root, self.size, self.num, self.train = args.dataset_dir, args.img_size, int(args.num_classes), train
        self.transform_ = transform
        is_train = {i: flag for i, flag in _pairs(os.path.join(self.root, "train_test_split.txt"))}
        for i, flag in is_train.items():
            if flag not in ("0", "1"):
                raise Exception("label error")
        label_of = dict(_pairs(os.path.join(self.root, "image_class_labels.txt")))
        self._train_path_label, self._test_path_label = [], []
        for image_id, image_name in _pairs(os.path.join(self.root, "images.txt")):
            if int(image_name[:3]) > self.num:                     # keep the first `num_classes` species only
                continue
            (self._train_path_label if is_train[image_id] == "1" else self._test_path_label).append(
                (image_name, label_of[image_id]))

    def _items(self):
        return self._train_path_label if self.train else self._test_path_label

    def __getitem__(self, index):
        image_name, label = self._items()[index]
        image_path = os.path.join(self.root, "images", image_name)
        img = Image.open(image_path)
        if img.mode != "RGB":                # reference converts 'L'; palette / alpha files would break its ToTensor
            img = img.convert("RGB")
        if self.transform_ is not None:
            img = self.transform_(img)
        return {"image": img, "label": torch.from_numpy(np.array(int(label) - 1)), "names": image_path}

    def __len__(self):
        return len(self._items())
