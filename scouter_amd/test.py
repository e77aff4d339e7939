"""Single-image inference + slot attention maps -- the `--vis` path of the reference (test.py:18-122 together with
sloter/utils/slot_attention.py:68-85): eval-mode forward of ONE image on the HIP path, per-class uint8 attention
maps written as `<out>/slot_{c}.png`, the input as `<out>/image.png`, and (optionally) the attention-area ratio the
reference prints with --cal_area_size (test.py:40-44).  The jet-colormap overlay of test.py:33-38 is matplotlib
post-processing of these PNGs and is not reproduced.

    python -m scouter_amd.test --model resnest26d --dataset ImageNet --channel 2048 --num_classes 10 \
        --slots_per_class 1 --to_k_layer 3 --power 2 --use_slot true --vis true --pre_trained false \
        [--checkpoint saved_model/ImageNet_use_slot_checkpoint.pth] [--image some.jpg] --img_size 224"""
import argparse
import os

import numpy as np
import torch

from .sloter.slot_model import SlotModel
from .train import checkpoint_name, get_args_parser


def load_image(path, size, channels):
    """RGB (or grey) image -> normalised float tensor [c, size, size] (dataset/transform_func.py:101-124: resize to
    img_size, ToTensor, ImageNet mean/std)."""
    from PIL import Image
    img = Image.open(path).convert("L" if channels == 1 else "RGB").resize((size, size), Image.BILINEAR)
    a = np.asarray(img, dtype=np.float32) / 255.0
    if channels == 1:
        a = (a[None] - 0.1307) / 0.3081
    else:
        a = (a.transpose(2, 0, 1) - np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]) / \
            np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
    return img, torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


@torch.no_grad()
def run(args, model, image, out_dir, label=None):
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    model.eval()
    output = model(image.unsqueeze(0).to(args.device, dtype=torch.float32))
    pred = int(output.argmax(dim=1))
    maps = model.slot.vis_maps()                      # [C, h, w] uint8, min-max scaled over the whole map (:74)
    for c, m in enumerate(maps):
        Image.fromarray(m, mode="L").save(os.path.join(out_dir, "slot_%d.png" % c))
    ratio = None
    if label is not None:
        idx = label if args.loss_status > 0 else min(label + 1, len(maps) - 1)
        ratio = float(maps[idx].astype(np.float64).sum()) / float(maps[idx].size * 255)
    return output[0].cpu(), pred, maps, ratio


def main():
    parser = argparse.ArgumentParser("single-image xSlot inference", parents=[get_args_parser()])
    parser.add_argument("--checkpoint", default="", help="checkpoint written by train.py (default: the standard name "
                                                         "under --output_dir, if it exists)")
    parser.add_argument("--image", default="", help="image file; a seeded synthetic image is used when empty")
    parser.add_argument("--label", default=None, type=int)
    parser.add_argument("--vis_dir", default="sloter/vis")
    args = parser.parse_args()
    for name, typ in (("num_classes", int), ("lambda_value", float), ("power", int), ("slots_per_class", int)):
        setattr(args, name, typ(getattr(args, name)))
    args.vis, args.vis_id = False, 0                  # maps are taken from model.slot.last_attn, image index 0
    model = SlotModel(args).to(args.device)
    ckpt = args.checkpoint or os.path.join(args.output_dir, checkpoint_name(args))
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=False)["model"], strict=True)
        print("load", ckpt)
    else:
        print("no checkpoint found (%s): random weights" % ckpt)
    channels, size = (1 if args.dataset == "MNIST" else 3), int(args.img_size)
    if args.image:
        raw, image = load_image(args.image, size, channels)
        os.makedirs(args.vis_dir, exist_ok=True)
        raw.save(os.path.join(args.vis_dir, "image.png"))
    else:
        image = torch.from_numpy(np.random.default_rng(0).standard_normal((channels, size, size), dtype=np.float32))
    out, pred, maps, ratio = run(args, model, image, args.vis_dir, args.label)
    print(out)
    print(pred)
    if ratio is not None and args.cal_area_size:
        print(f"attention_ratio: {ratio}")


if __name__ == "__main__":
    main()
