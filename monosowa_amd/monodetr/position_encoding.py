"""Sine positional encoding (reference: lib/models/monodetr/position_encoding.py:19-56,
build at :88-99: ``PositionEmbeddingSine(hidden_dim // 2, normalize=True)``)."""
import math

import torch
from torch import nn

from .misc import NestedTensor


CACHE_ALL_VALID = True


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor):
        x, mask = tensor_list.tensors, tensor_list.mask
        assert mask is not None
        # with an all-False mask (guaranteed by construction when all_valid) the encoding depends on the shape only:
        # computed once per (B, H, W, device) instead of ~20 small kernels per level and step
        if CACHE_ALL_VALID and getattr(tensor_list, "all_valid", False):
            key = (tuple(mask.shape), x.device)
            cache = self.__dict__.setdefault("_cache", {})
            if key not in cache:
                cache[key] = self._encode(x, mask).detach()
            return cache[key]
        return self._encode(x, mask)

    def _encode(self, x, mask):
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def build_position_encoding(cfg):
    if cfg["position_embedding"] in ("v2", "sine"):
        return PositionEmbeddingSine(cfg["hidden_dim"] // 2, normalize=True)
    raise ValueError("not supported %s (only the sine encoding is on the shipped path)" % cfg["position_embedding"])
