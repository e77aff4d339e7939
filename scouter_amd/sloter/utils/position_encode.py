"""Sine positional encoding -- mirrors sloter/utils/position_encode.py:10-46,77-87 of the reference.
The table is a constant of the (h, w, d) grid: it is computed once per device by scouter_posenc_sine_f32 (the
reference rebuilds it from ~20 small ops every forward)."""
import math

import torch
from torch import nn

from ... import kernels as K


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if not normalize or temperature != 10000 or (scale is not None and abs(scale - 2 * math.pi) > 1e-12):
            raise NotImplementedError("the HIP table kernel implements the configuration SCOUTER uses: "
                                      "normalize=True, temperature=1e4, scale=2*pi")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi

    def table(self, h, w, device):
        """Token-major [h*w, 2*num_pos_feats] table (what the fused xSlot kernel consumes)."""
        return K.posenc_sine(h, w, 2 * self.num_pos_feats, device)

    def forward(self, x):
        """x: [B, C, h, w] -> pos [B, 2*num_pos_feats, h, w] (reference signature, position_encode.py:26-46)."""
        b, c, h, w = x.shape
        pe = self.table(h, w, x.device).t().reshape(1, -1, h, w)
        return pe.expand(b, -1, -1, -1).to(x.dtype)


def build_position_encoding(position_embedding, hidden_dim):
    if position_embedding in ("v2", "sine"):
        return PositionEmbeddingSine(hidden_dim // 2, normalize=True)
    raise ValueError("not supported %s (the xSlot path uses the sine encoding)" % position_embedding)
