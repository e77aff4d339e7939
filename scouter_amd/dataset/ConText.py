"""Folder-based readers (reference dataset/ConText.py): `MakeList` (ConText: one folder, class = file-name prefix up to
the first '_', 80/20 split with sklearn's train_test_split(random_state=1)), `MakeListImage` (ImageNet layout:
train/<wnid>/*, val/<wnid>/*, the first `num_classes` folders in sorted order) and the `ConText` Dataset over such a
[path, label] list.  Images are opened as RGB."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from ..tools.prepare_things import get_name


class MakeList(object):
    def __init__(self, args, ratio=0.8):
        self.image_root = args.dataset_dir
        self.all_image = get_name(self.image_root, mode_folder=False)
        self.category = sorted({name[:name.find("_")] for name in self.all_image})
        for c_id, c in enumerate(self.category):
            print(c_id, "\t", c)
        self.ration = ratio

    def deal_label(self, img_name):
        return self.category.index(img_name[:img_name.find("_")])

    def get_data(self):
        from sklearn.model_selection import train_test_split
        items = [[os.path.join(self.image_root, name), self.deal_label(name)] for name in self.all_image]
        return train_test_split(items, random_state=1, train_size=self.ration)


class MakeListImage(object):
    def __init__(self, args):
        self.image_root = args.dataset_dir
        self.category = get_name(os.path.join(self.image_root, "train"))
        self.used_cat = self.category[:int(args.num_classes)]

    def deal_label(self, folder):
        return self.used_cat.index(folder)

    def get_img(self, folders, phase):
        out = []
        for folder in folders:
            root = os.path.join(self.image_root, phase, folder)
            out += [[os.path.join(root, name), self.deal_label(folder)] for name in get_name(root, mode_folder=False)]
        return out

    def get_data(self):
        return self.get_img(self.used_cat, "train"), self.get_img(self.used_cat, "val")


class ConText(Dataset):
    def __init__(self, data, transform=None):
        self.all_item, self.transform = data, transform

    def __len__(self):
        return len(self.all_item)

    def __getitem__(self, item_id):
        image_path, label = self.all_item[item_id][0], self.all_item[item_id][1]
        if not os.path.exists(image_path):
            raise FileNotFoundError("not exist image:" + image_path)
        image = Image.open(image_path).convert("RGB")
        if self.transform:
            image = self.transform(image)
        return {"image": image, "label": torch.from_numpy(np.array(label)), "names": image_path}
