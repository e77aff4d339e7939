"""create_model for the backbones of the xSlot hot path (reference: timm/models/factory.py:6-67)."""
from . import resnest, resnet

_MODELS = {"resnet18": resnet.resnet18, "resnest26d": resnest.resnest26d, "resnest50d": resnest.resnest50d}


def is_model(name):
    return name in _MODELS


def create_model(model_name, pretrained=False, num_classes=1000, in_chans=3, checkpoint_path="", **kwargs):
    if model_name not in _MODELS:
        raise RuntimeError("Unknown model (%s): scouter_amd builds resnet18 / resnest26d / resnest50d "
                           "(the backbones BASELINE.json names); the rest of the timm zoo is out of scope" % model_name)
    model = _MODELS[model_name](pretrained=pretrained, num_classes=num_classes, in_chans=in_chans, **kwargs)
    if checkpoint_path:
        import torch
        model.load_state_dict(torch.load(checkpoint_path, map_location="cpu", weights_only=False))
    return model
