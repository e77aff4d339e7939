"""TEST INFRASTRUCTURE ONLY -- harness that imports the *reference* (wbw520/scouter) read-only.

Runs ONLY in the build container (``/root/reference`` does not exist on the GPU box).  It is used by
``oracle/gen_golden.py`` to produce the committed fixtures under ``tests/golden/`` and by
``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent) to pin the restatement in
``oracle/torch_oracle.py`` against the real thing.  Nothing in ``scouter_amd/`` may import this module.

The reference needs three environment shims on torch 2.10; none modifies a reference file:
  1. stub ``torchvision`` (+ ``.transforms``, ``.transforms.functional``): ``timm/data/transforms.py:2`` imports it;
  2. stub ``torch._six`` (removed in torch>=2.0): ``timm/models/layers/helpers.py:6`` imports it;
  3. ``torch.normal(mu, sigma)`` with a negative-entry ``sigma`` tensor raises on modern torch
     (``sloter/utils/slot_attention.py:25``); while a model is being constructed ``sigma`` is ``abs()``-ed.
"""
import argparse
import collections.abc
import contextlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("SCOUTER_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "sloter", "slot_model.py"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _importable(name):
    import importlib.util
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def _tv_resize(img, size, interpolation=2):
    """torchvision.transforms.functional.resize for a PIL image and an (h, w) size: `img.resize((w, h), interp)` --
    the work is Pillow's (torchvision/transforms/functional_pil.py `resize`)."""
    if isinstance(size, int):
        raise NotImplementedError("the reference only passes (h, w) tuples (transform_func.py:115,122)")
    return img.resize((size[1], size[0]), interpolation)


def _tv_normalize(tensor, mean, std, inplace=False):
    """torchvision.transforms.functional.normalize: (x - mean[:, None, None]) / std[:, None, None] in x's dtype."""
    mean = torch.as_tensor(mean, dtype=tensor.dtype)[:, None, None]
    std = torch.as_tensor(std, dtype=tensor.dtype)[:, None, None]
    return tensor.clone().sub_(mean).div_(std)


def install_shims():
    """Idempotent: registers the stub modules and puts the reference on sys.path."""
    if "torchvision" not in sys.modules:
        class _Dummy:
            def __init__(self, *a, **k):
                pass
        names = ["Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomHorizontalFlip",
                 "RandomVerticalFlip", "ColorJitter", "RandomResizedCrop"]
        tv = _stub("torchvision")
        tr = _stub("torchvision.transforms", **{n: _Dummy for n in names})
        trf = _stub("torchvision.transforms.functional", resize=_tv_resize, normalize=_tv_normalize)
        tv.transforms = tr
        tr.functional = trf
    if "tools.image_aug" not in sys.modules and not _importable("imgaug"):
        # dataset/transform_func.py:2 imports the imgaug pipeline at module level; only `--aug true` would use it
        _stub("tools.image_aug", ImageAugment=None)
    if "torch._six" not in sys.modules:
        _stub("torch._six", container_abcs=collections.abc, string_classes=(str, bytes), int_classes=int)
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


@contextlib.contextmanager
def abs_sigma_normal():
    """torch 1.6 accepted a signed std tensor in torch.normal; torch 2.x does not (shim 3)."""
    orig = torch.normal

    def patched(mean, std, *a, **k):
        if isinstance(std, torch.Tensor):
            std = std.abs()
        return orig(mean, std, *a, **k)

    torch.normal = patched
    try:
        yield
    finally:
        torch.normal = orig


def make_args(model="resnest26d", num_classes=10, slots_per_class=1, channel=2048, to_k_layer=3, power=2,
              loss_status=1, lambda_value="1", dataset="ImageNet", use_slot=True, hidden_dim=64,
              freeze_layers=0, pre_trained=False, vis=False, vis_id=0):
    """The ``args`` fields ``SlotModel``/``load_backbone`` consume (reference train.py:28-66)."""
    return argparse.Namespace(
        model=model, pre_trained=pre_trained, num_classes=num_classes, dataset=dataset, use_slot=use_slot,
        use_pre=False, grad=False, channel=channel, slots_per_class=slots_per_class, hidden_dim=hidden_dim,
        freeze_layers=freeze_layers, vis=vis, vis_id=vis_id, loss_status=loss_status, power=power,
        to_k_layer=to_k_layer, lambda_value=lambda_value)


def build_reference_slot_model(args, feature_size=None):
    install_shims()
    from sloter.slot_model import SlotModel  # noqa: reference import
    with abs_sigma_normal():
        m = SlotModel(args)
    if feature_size is not None and args.use_slot:
        m.feature_size = feature_size  # reference hard-codes 9 (260x260 inputs), slot_model.py:64
    return m


def build_reference_slot_attention(num_classes, slots_per_class, dim, **kw):
    install_shims()
    from sloter.utils.slot_attention import SlotAttention  # noqa: reference import
    with abs_sigma_normal():
        return SlotAttention(num_classes, slots_per_class, dim, **kw)


@contextlib.contextmanager
def capture_python_sigmoid(store):
    """Records the outputs of python-level torch.sigmoid calls (the xSlot attention maps,
    slot_attention.py:57); nn.GRU's internal sigmoid is C++ and is not intercepted."""
    orig = torch.sigmoid

    def patched(x):
        y = orig(x)
        store.append(y.detach().clone())
        return y

    torch.sigmoid = patched
    try:
        yield
    finally:
        torch.sigmoid = orig
