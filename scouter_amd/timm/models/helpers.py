"""Pretrained-weight loading for the three backbones of the xSlot path (reference: timm/models/helpers.py:68-101
`load_pretrained`, called by the model entry points resnet.py:516-521 / resnest.py:165-189 when
`create_model(..., pretrained=True)`).

The reference downloads `default_cfg['url']` through `torch.utils.model_zoo.load_url`, which caches the file as
`$TORCH_HOME/hub/checkpoints/<basename of the url>`.  This build never touches the network: the same file is looked
up in `$SCOUTER_PRETRAINED_DIR`, then in torch hub's cache directory (where a machine that has run the reference
already holds it).  When it is missing the call RAISES -- `pre_trained=True` also freezes `freeze_layers` stages
(sloter/slot_model.py:79-94), and freezing randomly initialised layers would silently train a different model."""
import os

import torch

PRETRAINED = {   # model -> (file name, url of the reference's default_cfg, classes of the checkpoint's classifier)
    "resnet18": ("resnet18-5c106cde.pth", "https://download.pytorch.org/models/resnet18-5c106cde.pth", 1000),
    "resnest26d": ("gluon_resnest26-50eb607c.pth", "https://github.com/rwightman/pytorch-image-models/releases/download/"
                   "v0.1-weights/gluon_resnest26-50eb607c.pth", 1000),
    "resnest50d": ("resnest50-528c19ca.pth", "https://github.com/rwightman/pytorch-image-models/releases/download/"
                   "v0.1-resnest/resnest50-528c19ca.pth", 1000),
}


def pretrained_search_dirs():
    dirs = []
    if os.environ.get("SCOUTER_PRETRAINED_DIR"):
        dirs.append(os.environ["SCOUTER_PRETRAINED_DIR"])
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")), "hub",
                       "checkpoints")
    dirs.append(hub)
    return dirs


def find_pretrained(model_name):
    fname = PRETRAINED[model_name][0]
    for d in pretrained_search_dirs():
        path = os.path.join(d, fname)
        if os.path.isfile(path):
            return path
    return None


def load_pretrained(model, model_name, num_classes=1000, in_chans=3, strict=True):
    """State-dict surgery as in the reference (helpers.py:77-101): 1-channel first conv = channel sum of the RGB
    filter; classifier dropped (and strict off) when the class count differs from the checkpoint's."""
    fname, url, ck_classes = PRETRAINED[model_name]
    path = find_pretrained(model_name)
    if path is None:
        raise FileNotFoundError(
            "pretrained weights for %s not found: the reference downloads %s; there is no network here -- put %s into "
            "$SCOUTER_PRETRAINED_DIR (searched: %s), or pass --pre_trained false (which also disables --freeze_layers)"
            % (model_name, url, fname, ", ".join(pretrained_search_dirs())))
    state = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(state, dict) and "state_dict" in state and not any(k.endswith(".weight") for k in state):
        state = state["state_dict"]
    state = dict(state)
    if in_chans == 1:
        first = "conv1.0.weight" if "conv1.0.weight" in state else "conv1.weight"
        state[first] = state[first].sum(dim=1, keepdim=True)
    elif in_chans != 3:
        raise ValueError("Invalid in_chans for pretrained weights")
    if num_classes != ck_classes:
        state.pop("fc.weight", None)
        state.pop("fc.bias", None)
        strict = False
    missing, unexpected = model.load_state_dict(state, strict=False)
    # checkpoints saved before torch 0.4.1 (the genuine resnet18-5c106cde.pth has 102 keys) carry no
    # `num_batches_tracked`; torch's own BatchNorm fills the absent counter in silently (_NormBase._load_from_state_dict,
    # version < 2) and so does this loader: the buffer keeps its initial 0
    missing = [k for k in missing if not k.endswith(".num_batches_tracked")]
    missing = [k for k in missing if not (k.startswith("fc.") and not strict)]
    if missing or unexpected:
        raise RuntimeError("pretrained checkpoint %s does not fit %s: missing %s, unexpected %s"
                           % (path, model_name, missing[:6], list(unexpected)[:6]))
    return path
