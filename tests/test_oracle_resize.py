"""Input pipeline, CPU side: the Pillow-resampler restatement (oracle/pil_resize.py) is pinned to (a) the fixtures the
REFERENCE's make_transform produced (tests/golden/resize_cases.npz, oracle/gen_golden_resize.py) and (b) Pillow itself
when it is importable; the dataset readers and the host half of the transform are exercised on generated files."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import pil_resize as P
from oracle.gen_golden_resize import CASES

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_cases.npz"))


@pytest.mark.parametrize("case", [c[0] for c in CASES])
def test_oracle_matches_reference_fixture(case):
    size, c = (int(v) for v in GOLD[case + "_meta"])
    img = GOLD[case + "_in"]
    res = P.resize_bilinear_u8(img if c == 3 else img[:, :, 0], size, size)
    res = res if res.ndim == 3 else res[:, :, None]
    assert np.array_equal(res, GOLD[case + "_resized"])                       # bytes: bit-exact
    out = P.transform(img if c == 3 else img[:, :, 0], size, str(GOLD[case + "_dataset"]))
    assert np.array_equal(out, GOLD[case + "_out"])                           # float32 values: bit-exact


def test_oracle_matches_pillow_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for _ in range(25):
        h, w, oh, ow = (int(v) for v in rng.integers(3, 70, 4))
        c = int(rng.choice([1, 3]))
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        pil = Image.fromarray(a[:, :, 0], mode="L") if c == 1 else Image.fromarray(a, mode="RGB")
        ref = np.array(pil.resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(P.resize_bilinear_u8(a[:, :, 0] if c == 1 else a, oh, ow), ref), (h, w, oh, ow, c)


def test_normalize_table_is_the_reference_arithmetic():
    from scouter_amd.dataset.transform_func import NORMALIZE_VALUE, normalize_table
    for name, (mean, std) in NORMALIZE_VALUE.items():
        assert P.NORMALIZE[name] == (mean, std)
        lut = normalize_table(mean, std).numpy()
        assert np.array_equal(lut, P.normalize_lut(mean, std))
        v = np.arange(256, dtype=np.uint8)[:, None, None].repeat(len(mean), 2)     # [256, 1, C] image
        assert np.array_equal(P.to_tensor_normalize(v, mean, std)[:, :, 0], lut)


def _args(**kw):
    base = dict(dataset="ImageNet", img_size=16, aug=False, num_classes=2, dataset_dir="", synthetic_data=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_readers_and_host_transform(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from scouter_amd.dataset.choose_dataset import select_dataset
    from scouter_amd.dataset.transform_func import collate_raw, make_transform
    rng = np.random.default_rng(3)
    # ---- ImageNet layout: train/<wnid>/*.png, val/<wnid>/*.png; only the first num_classes folders are used
    for phase, n in (("train", 3), ("val", 2)):
        for wnid in ("n01", "n02", "n03"):
            d = tmp_path / "inet" / phase / wnid
            d.mkdir(parents=True)
            for k in range(n):
                Image.fromarray(rng.integers(0, 256, (9 + k, 7 + 2 * k, 3), dtype=np.uint8)).save(d / ("%d.png" % k))
    tr, va = select_dataset(_args(dataset_dir=str(tmp_path / "inet") + "/"))
    assert len(tr) == 6 and len(va) == 4
    s = tr[4]
    assert s["image"].dtype == torch.uint8 and s["image"].shape == (10, 9, 3) and int(s["label"]) == 1
    batch = collate_raw([tr[0], tr[5]])
    from scouter_amd.dataset.transform_func import PackedImages
    # (round 4: the frames of a batch travel packed in ONE buffer -- one pinned allocation, one H2D copy -- indexable like a list)
    assert isinstance(batch["image"], PackedImages) and len(batch["image"]) == 2
    assert torch.equal(batch["image"][0], tr[0]["image"]) and torch.equal(batch["image"][1], tr[5]["image"])
    assert batch["label"].tolist() == [0, 1] and batch["label"].dtype == torch.int64
    # ---- CUB-200 layout
    root = tmp_path / "cub"
    (root / "images" / "001.a").mkdir(parents=True)
    (root / "images" / "002.b").mkdir(parents=True)
    (root / "images" / "003.c").mkdir(parents=True)
    names = ["001.a/x.png", "001.a/y.png", "002.b/z.png", "003.c/w.png"]
    for i, nme in enumerate(names):
        mode_l = i == 1
        arr = rng.integers(0, 256, (8, 6) if mode_l else (8, 6, 3), dtype=np.uint8)
        Image.fromarray(arr).save(root / "images" / nme)
    (root / "images.txt").write_text("".join("%d %s\n" % (i + 1, n) for i, n in enumerate(names)))
    (root / "image_class_labels.txt").write_text("1 1\n2 1\n3 2\n4 3\n")
    (root / "train_test_split.txt").write_text("1 1\n2 0\n3 1\n4 1\n")
    tr, va = select_dataset(_args(dataset="CUB200", dataset_dir=str(root), num_classes=2))
    assert len(tr) == 2 and len(va) == 1                                       # class 3 filtered out
    assert va[0]["image"].shape == (8, 6, 3) and int(va[0]["label"]) == 0      # grayscale file promoted to RGB
    assert int(tr[1]["label"]) == 1 and tr[1]["names"].endswith("002.b/z.png")
    # ---- MNIST from raw IDX files
    raw = tmp_path / "mn" / "MNIST" / "raw"
    raw.mkdir(parents=True)
    imgs = rng.integers(0, 256, (5, 28, 28), dtype=np.uint8)
    labs = np.array([3, 1, 4, 1, 5], np.uint8)
    for stem in ("train", "t10k"):
        (raw / (stem + "-images-idx3-ubyte")).write_bytes(b"\x00\x00\x08\x03" + np.array([5, 28, 28], ">u4").tobytes() + imgs.tobytes())
        (raw / (stem + "-labels-idx1-ubyte")).write_bytes(b"\x00\x00\x08\x01" + np.array([5], ">u4").tobytes() + labs.tobytes())
    tr, va = select_dataset(_args(dataset="MNIST", dataset_dir=str(tmp_path / "mn")))
    assert len(tr) == 5 and tr[2]["label"] == 4 and tr[2]["image"].shape == (28, 28, 1)
    assert np.array_equal(tr[2]["image"][:, :, 0].numpy(), imgs[2])
    # ---- aug is refused, unknown mode raises like the reference
    with pytest.raises(NotImplementedError):
        make_transform(_args(aug=True), "train")
    with pytest.raises(ValueError):
        make_transform(_args(), "test")
