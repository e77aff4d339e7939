import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, whatever -m says."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def pretrained_dir(tmp_path, monkeypatch):
    """Factory: writes a synthetic timm-format ImageNet checkpoint (1000-class classifier, reference key names, dense
    OIHW tensors) for `arch` under the name the reference would download, and points SCOUTER_PRETRAINED_DIR at it."""
    import torch
    d = tmp_path / "pretrained"
    d.mkdir()
    monkeypatch.setenv("SCOUTER_PRETRAINED_DIR", str(d))
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "no_torch_home"))

    def make(arch, seed=11, legacy=False):
        """legacy=True: no `num_batches_tracked` entries -- the genuine torchvision resnet18-5c106cde.pth (102 keys)
        was saved before torch 0.4.1 introduced that buffer."""
        from scouter_amd.timm.models import create_model
        from scouter_amd.timm.models.helpers import PRETRAINED
        g = torch.Generator().manual_seed(seed)
        m = create_model(arch, pretrained=False, num_classes=1000)
        sd = {}
        for k, v in m.state_dict().items():
            sd[k] = (torch.randn(v.shape, generator=g) * 0.05 if v.dtype.is_floating_point else v.clone()).contiguous()
            if k.endswith("running_var"):
                sd[k] = sd[k].abs() + 0.5
        if legacy:
            sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
        torch.save(sd, d / PRETRAINED[arch][0])
        return sd
    return make
