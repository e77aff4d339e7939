from .split_attn import SplitAttnConv2d  # noqa: F401
