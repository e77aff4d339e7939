"""Image pre-processing of the reference (dataset/transform_func.py) split where the hardware wants it:

    host (DataLoader workers)   decode only: PIL image -> uint8 [h, w, c] array at its NATIVE size (`Decode`)
    GPU (one batch, 2 launches) Resize((S, S)) + ToTensor + Normalize  (`GpuTransform` -> kernels.resize_normalize)

The reference resizes, converts and normalises every image on the CPU in float64 (transform_func.py:19-31, 51-66,
91-99) -- at MI355X step rates (thousands of images / s) that is the bottleneck of real-data training.  The GPU path is
bit-identical to it: the resize reproduces Pillow's 8-bit bilinear resampler exactly and ToTensor + Normalize become a
256-entry table per channel built here in float64 the way the reference computes the values (then cast to float32, as
engine.py:25 does).  `--aug true` (imgaug pipeline, tools/image_aug.py) is host-side augmentation with a third-party
package and is refused."""
import numpy as np
import torch

NORMALIZE_VALUE = {"MNIST": [[0.1307], [0.3081]],                                   # transform_func.py:102-105
                   "CUB200": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]],
                   "ConText": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]],
                   "ImageNet": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]}


class Decode(object):
    """PIL image -> contiguous uint8 [h, w, c] tensor (c = 1 for mode 'L'); runs in the DataLoader workers."""

    def __call__(self, image):
        a = np.asarray(image, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.array(a, dtype=np.uint8, order="C"))      # own, writable copy

    def __repr__(self):
        return self.__class__.__name__ + "()"


def normalize_table(mean, std):
    """[C, 256] float32: (v / 255 - mean) / std evaluated in float64 (ToTensor then Normalize), for every byte v."""
    v = np.arange(256, dtype=np.float64)[None, :] / 255
    m, s = np.asarray(mean, np.float64)[:, None], np.asarray(std, np.float64)[:, None]
    return torch.from_numpy(((v - m) / s).astype(np.float32))


class GpuTransform(object):
    """Batch-level Resize + ToTensor + Normalize on the device (the `val` / aug-free `train` transform)."""

    def __init__(self, dataset, img_size):
        mean, std = NORMALIZE_VALUE[dataset]
        self.size = int(img_size)
        self.table = normalize_table(mean, std)
        self._dev_table = {}

    def __call__(self, images_u8, device):
        from .. import kernels
        key = str(device)
        if key not in self._dev_table:
            self._dev_table[key] = self.table.to(device)
        dev_imgs = [t.to(device, non_blocking=True) for t in images_u8]
        return kernels.resize_normalize(dev_imgs, self.size, self._dev_table[key])

    def __repr__(self):
        return "GpuTransform(size=%d)" % self.size


def make_transform(args, mode):
    """Host half of the transform (what the Dataset applies per sample).  The device half is `make_gpu_transform`."""
    if mode not in ("train", "val"):
        raise ValueError(f"unknown {mode}")
    if mode == "train" and getattr(args, "aug", False):
        raise NotImplementedError("--aug true (imgaug, tools/image_aug.py) is host-side augmentation outside the "
                                  "xSlot hot path")
    return Decode()


def make_gpu_transform(args):
    return GpuTransform(args.dataset, args.img_size)


def collate_raw(samples):
    """Keeps the decoded images as a list (they differ in size); labels become one int64 tensor."""
    return {"image": [s["image"] for s in samples],
            "label": torch.as_tensor([int(s["label"]) for s in samples], dtype=torch.int64),
            "names": [s.get("names", "") for s in samples]}
