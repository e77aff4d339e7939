from .factory import create_model  # noqa: F401
from .resnet import ResNet, BasicBlock, resnet18  # noqa: F401
from .resnest import ResNestBottleneck, resnest26d, resnest50d  # noqa: F401
