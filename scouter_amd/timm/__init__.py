"""The three backbones of the xSlot hot path (resnet18, resnest26d, resnest50d) with the reference's timm module
and parameter names, built from the explicit-forward/backward HIP layers of scouter_amd.nn_hip.
Mirrors: timm/models/{resnet,resnest,factory}.py and timm/models/layers/split_attn.py of the reference."""
from .models import create_model  # noqa: F401
