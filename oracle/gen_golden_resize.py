"""TEST INFRASTRUCTURE.  Generates tests/golden/resize_cases.npz with the REFERENCE's own pre-processing pipeline:
`dataset.transform_func.make_transform(args, "val")` imported from /root/reference (Resize -> ToTensor -> Normalize,
transform_func.py:101-124) running on Pillow (this image: see the `pillow_version` field).  torchvision is absent, so
its two one-line glue functions (`F.resize` -> `PIL.Image.resize`, `F.normalize`) come from oracle/ref_import.py.

    python oracle/gen_golden_resize.py

Cases: up- and down-scaling, odd sizes, 1 and 3 channels, identity in one dimension, a 20x down-scale."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import  # noqa: E402

CASES = [  # (name, dataset, in_h, in_w, out_size)
    ("down_rgb", "ImageNet", 37, 53, 16), ("up_rgb", "CUB200", 9, 7, 24), ("mixed_rgb", "ConText", 40, 11, 20),
    ("same_w_rgb", "ImageNet", 31, 18, 18), ("mnist_up", "MNIST", 28, 28, 64), ("mnist_down", "MNIST", 28, 28, 12),
    ("big_down_rgb", "ImageNet", 330, 41, 16), ("odd_rgb", "ImageNet", 17, 129, 33),
]


def main():
    ref_import.install_shims()
    import PIL
    from PIL import Image
    from dataset.transform_func import make_transform      # the reference's
    out = {"pillow_version": np.array(PIL.__version__)}
    rng = np.random.default_rng(20240928)
    for name, dataset, h, w, size in CASES:
        c = 1 if dataset == "MNIST" else 3
        # smooth + noisy content so that rounding ties and saturation both occur
        base = rng.integers(0, 256, (h, w, c)).astype(np.float64)
        ramp = np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, c))
        img = np.clip(0.5 * base + 0.5 * ramp + rng.normal(0, 40, (h, w, c)), 0, 255).astype(np.uint8)
        args = type("A", (), {"dataset": dataset, "img_size": size, "aug": False})()
        pil = Image.fromarray(img[:, :, 0], mode="L") if c == 1 else Image.fromarray(img, mode="RGB")
        ref = make_transform(args, "val")(pil)               # float64 tensor [c, size, size]
        resized = np.array(pil.resize((size, size), Image.BILINEAR))
        out[name + "_in"] = img
        out[name + "_resized"] = resized if resized.ndim == 3 else resized[:, :, None]
        out[name + "_out"] = ref.numpy().astype(np.float32)          # engine.py:25 casts to float32
        out[name + "_meta"] = np.array([size, c])
        out[name + "_dataset"] = np.array(dataset)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resize_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
