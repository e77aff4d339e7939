"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/scouter_hip.h declares, the
module surface mirrors the reference (state_dict keys / shapes, constructor args), conv weights keep the HWIO
physical layout through load_state_dict/.to(), and CPU tensors are refused (no fallback)."""
import argparse
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(model="resnest26d", C=10, spc=1, L=3, mnist=False, **kw):
    d = dict(model=model, pre_trained=False, num_classes=C, dataset="MNIST" if mnist else "ImageNet", use_slot=True,
             use_pre=False, grad=False, channel=O.ARCHS[model]["channel"], slots_per_class=spc, hidden_dim=64,
             freeze_layers=0, vis=False, vis_id=0, loss_status=1, power=2, to_k_layer=L, lambda_value="1")
    d.update(kw)
    return argparse.Namespace(**d)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from scouter_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = [n for n, _, _ in _native.declared_symbols()]
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert _native.lib().scouter_abi_version() == 1
    # the documents quote the number of entry points: keep them honest
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = re.search(r"declares (\d+) `extern \"C\"` entry points", open(os.path.join(root, "INTEGRATION.md")).read())
    assert m and int(m.group(1)) == len(names), (m and m.group(1), len(names))


@pytest.mark.parametrize("model,C,spc,L,mnist", [("resnet18", 10, 1, 1, True), ("resnest26d", 10, 1, 3, False),
                                                 ("resnest50d", 100, 3, 3, False)])
def test_state_dict_matches_reference_layout(model, C, spc, L, mnist):
    from scouter_amd.sloter.slot_model import SlotModel
    m = SlotModel(_args(model, C, spc, L, mnist))
    spec = O.state_dict_spec(model, C, spc, L, in_chans=1 if mnist else 3, mnist_stem=mnist)
    sd = m.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(spec[k]), k
    # round trip through a reference-layout state dict keeps values and the HWIO physical layout
    P = O.synth_state(spec, 3)
    m.load_state_dict(P)
    w = m.backbone.layer1[0].conv1.weight
    assert w.permute(2, 3, 1, 0).is_contiguous()
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), P[k]), k


def test_cpu_input_is_refused_not_emulated():
    from scouter_amd.sloter.slot_model import SlotModel
    m = SlotModel(_args("resnet18", mnist=True, L=1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 1, 64, 64), torch.tensor([0, 1]))


def test_freeze_layers_matches_reference_rule(pretrained_dir):
    from scouter_amd.sloter.slot_model import SlotModel
    sd = pretrained_dir("resnest26d")
    m = SlotModel(_args("resnest26d", pre_trained=True, freeze_layers=2))
    frozen = {n.split(".")[1] for n, p in m.named_parameters() if not p.requires_grad}
    assert frozen == {"conv1", "bn1", "layer1", "layer2"}
    # the frozen layers hold the PRETRAINED values (local file, reference helpers.py:68-101), fc discarded (10 != 1000)
    got = m.state_dict()
    for k, v in sd.items():
        if k.startswith("fc."):
            assert "backbone." + k not in got
        else:
            assert torch.equal(got["backbone." + k], v), k
    assert m.backbone.layer1[0].conv1.weight.permute(2, 3, 1, 0).is_contiguous()     # HWIO physical layout kept


def test_pre_trained_without_weights_raises_instead_of_freezing_random_layers(pretrained_dir):
    """`--pre_trained true` (the reference default) must not silently random-initialise and freeze."""
    from scouter_amd.sloter.slot_model import SlotModel
    with pytest.raises(FileNotFoundError, match="SCOUTER_PRETRAINED_DIR"):
        SlotModel(_args("resnet18", mnist=True, L=1, pre_trained=True, freeze_layers=2))
    with pytest.raises(FileNotFoundError, match="gluon_resnest26-50eb607c.pth"):
        SlotModel(_args("resnest26d", pre_trained=True, use_slot=False))


def test_pre_trained_mnist_swaps_the_stem_after_loading(pretrained_dir):
    """reference slot_model.py:19-24: 3-channel pretrained model first, THEN conv1 := fresh Conv2d(1, 64, 3, 2, 1)."""
    from scouter_amd.sloter.slot_model import SlotModel
    sd = pretrained_dir("resnet18")
    assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7)
    m = SlotModel(_args("resnet18", mnist=True, L=1, pre_trained=True, freeze_layers=1))
    assert tuple(m.backbone.conv1.weight.shape) == (64, 1, 3, 3)
    assert torch.equal(m.state_dict()["backbone.layer3.1.conv2.weight"], sd["layer3.1.conv2.weight"])
    assert not m.backbone.layer1[0].conv1.weight.requires_grad and m.backbone.layer2[0].conv1.weight.requires_grad


def test_pretrained_first_conv_channel_sum_for_one_channel_models(pretrained_dir):
    """helpers.py:77-81 of the reference: in_chans == 1 sums the RGB filter."""
    from scouter_amd.timm.models import create_model
    sd = pretrained_dir("resnet18")
    m = create_model("resnet18", pretrained=True, num_classes=1000, in_chans=1)
    assert torch.allclose(m.state_dict()["conv1.weight"], sd["conv1.weight"].sum(1, keepdim=True))
    assert torch.equal(m.state_dict()["fc.weight"], sd["fc.weight"])


def test_pretrained_legacy_checkpoint_without_num_batches_tracked_loads(pretrained_dir):
    """resnet18-5c106cde.pth predates `num_batches_tracked` (102 keys); torch's BatchNorm fills the counter in
    (_NormBase._load_from_state_dict), the reference therefore loads it, and so must this loader."""
    from scouter_amd.sloter.slot_model import SlotModel
    from scouter_amd.timm.models import create_model
    sd = pretrained_dir("resnet18", legacy=True)
    assert len(sd) == 102 and not any(k.endswith("num_batches_tracked") for k in sd)
    m = SlotModel(_args("resnet18", mnist=True, L=1, pre_trained=True, freeze_layers=1))
    got = m.state_dict()
    assert torch.equal(got["backbone.layer2.0.bn1.running_var"], sd["layer2.0.bn1.running_var"])
    assert int(got["backbone.layer2.0.bn1.num_batches_tracked"]) == 0
    # a genuinely missing tensor is still an error
    sd.pop("layer1.0.bn1.weight")
    from scouter_amd.timm.models.helpers import find_pretrained
    torch.save(sd, find_pretrained("resnet18"))
    with pytest.raises(RuntimeError, match="layer1.0.bn1.weight"):
        create_model("resnet18", pretrained=True, num_classes=1000)


def test_fused_adamw_plan_lookup_with_two_param_groups():
    """the plan of the second group was looked up with list.index (dict == dict compares tensors)."""
    from scouter_amd.optim import FusedAdamW
    a, b = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(4, 4))
    opt = FusedAdamW([{"params": [a]}, {"params": [b], "lr": 1e-2}], lr=1e-3)
    st = [{"step": 0}, {"step": 0}]
    opt.state["_flat_0"], opt.state["_flat_1"] = st
    a.grad, b.grad = torch.zeros(8), torch.zeros(4, 4)
    mk = lambda p, s: dict(table=torch.zeros(3), n=1, base=0, span=0, ids=[id(p)], gptrs=[p.grad.data_ptr()],
                           pptrs=[p.data_ptr()], state=s)
    opt._plan = [mk(a, st[0]), mk(b, st[1])]
    assert opt._plan_valid()                       # (raised "Boolean value of Tensor ... ambiguous" before)
    opt._plan[1]["state"] = {"step": 0}
    assert not opt._plan_valid()


def test_grad_arena_views_alias_flat_buffer():
    from scouter_amd.nn_hip import GradArena
    from scouter_amd.sloter.slot_model import SlotModel
    m = SlotModel(_args("resnet18", mnist=True, L=1))
    a = GradArena(m)
    assert "slot.to_q.0.weight" not in a.views
    total = sum((p.numel() + 3) // 4 * 4 for n, p in m.named_parameters() if "to_q" not in n)
    assert a.numel == total
    a.flat.copy_(torch.arange(a.flat.numel(), dtype=torch.float32))
    name, p, off, n = a.entries[0]
    assert a.views[name].shape == p.shape and a.views[name].stride() == p.stride()
    hw = a.views[name].permute(2, 3, 1, 0).reshape(-1)
    assert torch.equal(hw, a.flat[off:off + n])


def test_use_pre_stage2_handoff_loads_fc_baseline_checkpoint(tmp_path, monkeypatch):
    """README two-stage recipe (reference README.md:84-97, sloter/slot_model.py:26-33): the FC baseline (use_slot
    false) is saved under `saved_model/{dataset}_no_slot_checkpoint.pth`; the xSlot model built with use_pre=True
    strips `backbone.` (incl. the `backbone.fc.*` keys, which must exist at load time) and THEN replaces
    global_pool / fc by Identical."""
    from scouter_amd.sloter.slot_model import Identical, SlotModel
    from scouter_amd.tools import prepare_things as prt
    from scouter_amd.train import checkpoint_name
    monkeypatch.chdir(tmp_path)
    (tmp_path / "saved_model").mkdir()
    a1 = _args("resnet18", mnist=True, L=1, use_slot=False)
    a1.cal_area_size = False
    fc = SlotModel(a1)
    with torch.no_grad():                                    # make every tensor distinguishable from a fresh init
        for i, p in enumerate(fc.parameters()):
            p.add_(0.01 * (i + 1))
        fc.backbone.bn1.running_mean.fill_(0.25)
    assert checkpoint_name(a1) == "MNIST_no_slot_checkpoint.pth"
    prt.save_on_master({"model": fc.state_dict(), "epoch": 0}, tmp_path / "saved_model" / checkpoint_name(a1))
    assert "backbone.fc.weight" in fc.state_dict()
    m = SlotModel(_args("resnet18", mnist=True, L=1, use_slot=True, use_pre=True))
    assert isinstance(m.backbone.fc, Identical) and isinstance(m.backbone.global_pool, Identical)
    got, src = m.state_dict(), fc.state_dict()
    assert not any(k.startswith("backbone.fc.") for k in got)
    n = 0
    for k, v in src.items():
        if not k.startswith("backbone.fc."):
            assert torch.equal(got[k], v), k
            n += 1
    assert n == len(src) - 2
    assert m.backbone.layer1[0].conv1.weight.permute(2, 3, 1, 0).is_contiguous()
    # a checkpoint of another class count does not fit (strict load, like the reference)
    bad = SlotModel(_args("resnet18", C=7, mnist=True, L=1, use_slot=False))
    prt.save_on_master({"model": bad.state_dict()}, tmp_path / "saved_model" / "MNIST_no_slot_checkpoint.pth")
    with pytest.raises(RuntimeError, match="size mismatch"):
        SlotModel(_args("resnet18", mnist=True, L=1, use_slot=True, use_pre=True))


def test_packed_images_roundtrip_and_collate():
    """dataset.transform_func.PackedImages: frames of ragged sizes in one buffer at 16-byte aligned offsets, indexable
    like the list it replaces; collate_raw packs in the DataLoader worker (the H2D copy is then ONE per batch)."""
    import numpy as np
    from scouter_amd.dataset.transform_func import PackedImages, collate_raw
    rng = np.random.default_rng(3)
    frames = [torch.from_numpy(rng.integers(0, 256, (h, w, c), dtype=np.uint8))
              for h, w, c in ((5, 7, 3), (1, 1, 3), (33, 17, 3), (16, 16, 3))]
    p = PackedImages.pack(frames)
    assert len(p) == 4 and all(o % 16 == 0 for o in p.offsets) and p.flat.numel() == PackedImages.nbytes(p.shapes)
    for a, b in zip(frames, p):
        assert torch.equal(a, b)
    assert torch.equal(p[2], frames[2])
    big = torch.zeros(1 << 16, dtype=torch.uint8)
    q = PackedImages.pack(frames, out=big)                       # into a caller-owned (pinned staging) buffer
    assert q.flat.data_ptr() == big.data_ptr() and torch.equal(q[3], frames[3])
    with pytest.raises(RuntimeError):
        PackedImages.pack([frames[0].float()])
    batch = collate_raw([{"image": f, "label": i, "names": "n%d" % i} for i, f in enumerate(frames)])
    assert isinstance(batch["image"], PackedImages) and batch["label"].tolist() == [0, 1, 2, 3]
    assert torch.equal(batch["image"][1], frames[1])
    # travels through a worker process like any batch
    loader = torch.utils.data.DataLoader([{"image": f, "label": i} for i, f in enumerate(frames)], batch_size=2,
                                         collate_fn=collate_raw, num_workers=1)
    got = [b["image"] for b in loader]
    assert torch.equal(got[1][0], frames[2]) and got[0].shapes == [(5, 7, 3), (1, 1, 3)]


def test_device_batches_plain_iteration_on_cpu():
    """engine.device_batches on a CPU device (plumbing): every batch once, in order, already-transformed tensors pass
    through, labels become int64"""
    from scouter_amd import engine
    data = [{"image": torch.full((2, 1, 4, 4), float(i)), "label": torch.tensor([i, i + 1], dtype=torch.int32)} for i in range(3)]
    out = list(engine.device_batches(data, torch.device("cpu")))
    assert len(out) == 3
    for i, (x, y) in enumerate(out):
        assert x.dtype == torch.float32 and float(x[0, 0, 0, 0]) == i and y.dtype == torch.int64 and y.tolist() == [i, i + 1]
    assert list(engine.device_batches([], torch.device("cpu"))) == []
