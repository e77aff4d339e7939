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


class PackedImages(object):
    """A batch of decoded frames of different sizes as ONE contiguous uint8 buffer (frames back to back, each at a
    16-byte aligned offset) + their shapes: one pinned allocation and one host-to-device copy per batch instead of one
    pageable copy per frame (70 per step: 2.1 ms, 12 % of a training step, round 3).  Indexing / iteration give the frames
    back as [h, w, c] views, so code written for a list of frames keeps working.  DataLoader(pin_memory=True) pins it
    through `pin_memory()` in its background thread."""

    def __init__(self, flat, shapes):
        self.flat, self.shapes = flat, [tuple(int(v) for v in s) for s in shapes]
        self.offsets, off = [], 0
        for h, w, c in self.shapes:
            self.offsets.append(off)
            off += (h * w * c + 15) // 16 * 16
        assert off <= flat.numel()

    @staticmethod
    def nbytes(shapes):
        return sum((int(h) * int(w) * int(c) + 15) // 16 * 16 for h, w, c in shapes)

    @classmethod
    def pack(cls, images, out=None):
        """images: uint8 [h, w, c] host tensors -> PackedImages (in `out`, a flat uint8 buffer, when given and big enough)"""
        shapes = [tuple(t.shape) for t in images]
        n = cls.nbytes(shapes)
        flat = out[:n] if out is not None and out.numel() >= n else torch.empty(n, dtype=torch.uint8)
        p = cls(flat, shapes)
        for t, o, (h, w, c) in zip(images, p.offsets, shapes):
            if t.dtype != torch.uint8 or t.dim() != 3:
                raise RuntimeError("PackedImages: dense uint8 [h, w, c] frames expected, got %s %s" % (t.dtype, tuple(t.shape)))
            flat[o:o + h * w * c].copy_(t.reshape(-1))
        return p

    def pin_memory(self):
        return PackedImages(self.flat.pin_memory(), self.shapes)

    def frames(self, flat=None):
        flat = self.flat if flat is None else flat
        return [flat[o:o + h * w * c].view(h, w, c) for o, (h, w, c) in zip(self.offsets, self.shapes)]

    def __len__(self):
        return len(self.shapes)

    def __getitem__(self, i):
        h, w, c = self.shapes[i]
        return self.flat[self.offsets[i]:self.offsets[i] + h * w * c].view(h, w, c)

    def __iter__(self):
        return iter(self.frames())


class GpuTransform(object):
    """Batch-level Resize + ToTensor + Normalize on the device (the `val` / aug-free `train` transform).  The decoded
    frames travel as ONE pinned buffer in ONE asynchronous copy (PackedImages); the engine issues that copy and the two
    resize launches for batch n + 1 on a feed stream while step n computes (engine.device_batches)."""

    def __init__(self, dataset, img_size):
        mean, std = NORMALIZE_VALUE[dataset]
        self.size = int(img_size)
        self.table = normalize_table(mean, std)
        self._dev_table = {}
        self._staging = [None, None]          # pinned staging buffers for callers that hand over unpinned frames
        self._turn = 0
        self._in_flight = [None, None]        # event: the copy out of staging buffer i has finished

    def _stage(self, images):
        """list of frames / unpinned PackedImages -> PackedImages in one of two alternating pinned staging buffers"""
        shapes = images.shapes if isinstance(images, PackedImages) else [tuple(t.shape) for t in images]
        n = PackedImages.nbytes(shapes)
        i = self._turn
        self._turn ^= 1
        if self._in_flight[i] is not None:
            self._in_flight[i].synchronize()  # the copy issued two batches ago still reads this buffer (normally long done)
        buf = self._staging[i]
        if buf is None or buf.numel() < n:
            buf = self._staging[i] = torch.empty(max(n, 1 << 20) * 5 // 4, dtype=torch.uint8).pin_memory()
        if isinstance(images, PackedImages):
            buf[:n].copy_(images.flat[:n])
            return PackedImages(buf[:n], shapes), i
        return PackedImages.pack(list(images), out=buf), i

    def __call__(self, images_u8, device):
        from .. import kernels
        device = torch.device(device)
        key = str(device)
        if key not in self._dev_table:
            self._dev_table[key] = self.table.to(device)
        slot = None
        if isinstance(images_u8, PackedImages) and images_u8.flat.is_cuda:
            packed, dev_flat = images_u8, images_u8.flat
        else:
            packed = images_u8
            if not (isinstance(packed, PackedImages) and packed.flat.is_pinned()):
                packed, slot = self._stage(packed)
            dev_flat = torch.empty(packed.flat.numel(), dtype=torch.uint8, device=device)
            dev_flat.copy_(packed.flat, non_blocking=True)            # the ONE host-to-device copy of the batch
            if slot is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
                self._in_flight[slot] = ev
        return kernels.resize_normalize(packed.frames(dev_flat), self.size, self._dev_table[key])

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
    """Keeps the decoded images at their native sizes, packed into one buffer (PackedImages; indexable like a list);
    labels become one int64 tensor."""
    return {"image": PackedImages.pack([s["image"] for s in samples]),
            "label": torch.as_tensor([int(s["label"]) for s in samples], dtype=torch.int64),
            "names": [s.get("names", "") for s in samples]}
