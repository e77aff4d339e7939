"""Input pipeline on the GPU (C ABI scouter_resize_normalize_u8_f32): bit-exact against the fixtures produced by the
reference's make_transform, against the oracle on random batches of mixed sizes, and through the training engine."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import pil_resize as P
from oracle.gen_golden_resize import CASES

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_cases.npz"))


def _gpu_transform(dataset, size):
    from scouter_amd.dataset.transform_func import GpuTransform
    return GpuTransform(dataset, size)


@pytest.mark.parametrize("case", [c[0] for c in CASES])
def test_gpu_transform_matches_reference_fixture(case):
    size, c = (int(v) for v in GOLD[case + "_meta"])
    img = torch.from_numpy(GOLD[case + "_in"].copy())
    out = _gpu_transform(str(GOLD[case + "_dataset"]), size)([img], torch.device("cuda"))
    assert out.shape == (1, c, size, size) and out.dtype == torch.float32
    assert np.array_equal(out[0].cpu().numpy(), GOLD[case + "_out"])          # bit-exact


def test_gpu_transform_mixed_batch_matches_oracle():
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)
            for h, w in [(37, 91), (224, 224), (500, 375), (64, 300), (5, 4), (1100, 23), (96, 96)]]
    size = 96
    out = _gpu_transform("ImageNet", size)([torch.from_numpy(a) for a in imgs], torch.device("cuda")).cpu().numpy()
    for b, a in enumerate(imgs):
        assert np.array_equal(out[b], P.transform(a, size, "ImageNet")), "image %d (%s)" % (b, a.shape)


def test_too_large_downscale_is_refused():
    from scouter_amd import kernels as K
    lut = torch.zeros(3, 256, device="cuda")
    with pytest.raises(RuntimeError, match="down-scaling"):
        K.resize_normalize([torch.zeros(2000, 8, 3, dtype=torch.uint8, device="cuda")], 16, lut)


def test_engine_consumes_raw_batches():
    """one epoch of the engine over a loader that yields decoded uint8 images of different sizes"""
    from scouter_amd import engine
    from scouter_amd.dataset.transform_func import collate_raw, make_gpu_transform
    from scouter_amd.optim import FusedAdamW
    from scouter_amd.sloter.slot_model import SlotModel
    from scouter_amd.tools.calculate_tool import MetricLog
    from scouter_amd.tools.prepare_things import DataLoaderX
    from scouter_amd.train import get_args_parser
    args = get_args_parser().parse_args(["--dataset", "MNIST", "--model", "resnet18", "--channel", "512", "--img_size", "64",
                                         "--num_classes", "10", "--slots_per_class", "1", "--pre_trained", "false",
                                         "--batch_size", "4"])
    for name, typ in (("num_classes", int), ("lambda_value", float), ("power", int), ("slots_per_class", int)):
        setattr(args, name, typ(getattr(args, name)))
    rng = np.random.default_rng(5)

    class Raw(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            r = np.random.default_rng(100 + i)
            return {"image": torch.from_numpy(r.integers(0, 256, (20 + i, 28 + 2 * i, 1), dtype=np.uint8)),
                    "label": int(r.integers(0, 10))}

    loader = DataLoaderX(Raw(), batch_size=4, collate_fn=collate_raw)
    loader.gpu_transform = make_gpu_transform(args)
    torch.manual_seed(0)
    model = SlotModel(args).cuda()
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    log = MetricLog()
    engine.train_one_epoch(model, loader, opt, torch.device("cuda"), log.record, 0)
    engine.evaluate(model, loader, torch.device("cuda"), log.record, 0)
    assert np.isfinite(log.record["train"]["loss"][-1]) and np.isfinite(log.record["val"]["loss"][-1])
    # the images the model saw are exactly the reference's transform of the raw files
    batch = next(iter(loader))
    x = loader.gpu_transform(batch["image"], torch.device("cuda")).cpu().numpy()
    for b in range(4):
        assert np.array_equal(x[b], P.transform(batch["image"][b][:, :, 0].numpy(), 64, "MNIST"))


def test_train_main_on_image_folders(tmp_path):
    """`python -m scouter_amd.train --synthetic_data false --dataset ImageNet --dataset_dir <folders>` end to end"""
    Image = pytest.importorskip("PIL.Image")
    from scouter_amd import train
    rng = np.random.default_rng(9)
    for phase, n in (("train", 4), ("val", 2)):
        for wnid in ("n001", "n002"):
            d = tmp_path / "data" / phase / wnid
            d.mkdir(parents=True)
            for k in range(n):
                h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
                Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / ("%d.jpg" % k), quality=90)
    args = train.get_args_parser().parse_args([
        "--dataset", "ImageNet", "--model", "resnet18", "--channel", "512", "--img_size", "64", "--num_classes", "2",
        "--slots_per_class", "1", "--pre_trained", "false", "--batch_size", "4", "--epochs", "1", "--num_workers", "0",
        "--synthetic_data", "false", "--dataset_dir", str(tmp_path / "data") + "/", "--output_dir", ""])
    accs = train.param_translation(args)
    assert len(accs) == 2 and all(0.0 <= a <= 1.0 for a in accs)


def test_prefetching_feed_delivers_the_same_batches_as_the_direct_transform():
    """engine.device_batches: batch n + 1 is copied (one pinned buffer, one async H2D on the feed stream) and transformed
    while step n runs; what arrives on the compute stream is bit-identical to transforming each batch directly -- with
    DataLoader workers + pin_memory, and for unpinned batches through the transform's own pinned staging buffers (every
    buffer is poisoned / recycled in between: a missing event or record_stream would show)."""
    from scouter_amd import engine
    from scouter_amd.dataset.transform_func import GpuTransform, PackedImages, collate_raw
    from scouter_amd.tools.prepare_things import DataLoaderX

    class Raw(torch.utils.data.Dataset):
        def __len__(self):
            return 24

        def __getitem__(self, i):
            r = np.random.default_rng(500 + i)
            return {"image": torch.from_numpy(r.integers(0, 256, (40 + 3 * i, 64 - i, 3), dtype=np.uint8)),
                    "label": int(r.integers(0, 10))}
    dev = torch.device("cuda")
    tf = GpuTransform("ImageNet", 64)
    ds = Raw()
    want = []
    for b0 in range(0, 24, 4):
        frames = [ds[i]["image"] for i in range(b0, b0 + 4)]
        want.append((tf(frames, dev).clone(), [ds[i]["label"] for i in range(b0, b0 + 4)]))
    torch.cuda.synchronize()
    for kwargs in (dict(num_workers=2, pin_memory=True), dict(num_workers=0, pin_memory=False)):
        loader = DataLoaderX(ds, batch_size=4, collate_fn=collate_raw, **kwargs)
        loader.gpu_transform = GpuTransform("ImageNet", 64)
        n = 0
        for (x, y), (xw, yw) in zip(engine.device_batches(loader, dev), want):
            # the "training step": work on the compute stream that recycles memory while the next batch is being staged
            junk = [torch.full((1 << 18,), float("nan"), device=dev) for _ in range(8)]
            got = x.clone()
            del junk
            assert torch.equal(got, xw), (kwargs, n)
            assert y.tolist() == yw and y.dtype == torch.int64 and y.is_cuda
            n += 1
        assert n == 6
    # an already packed batch on the device goes straight to the kernels
    p = PackedImages.pack([ds[i]["image"] for i in range(4)])
    on_dev = PackedImages(p.flat.cuda(), p.shapes)
    assert torch.equal(tf(on_dev, dev), want[0][0])
