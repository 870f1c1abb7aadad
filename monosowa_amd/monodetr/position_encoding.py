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
        """Channel c of each half is a sinusoid of the (normalised) count of valid pixels up to and including this one, along
        y for the first half and along x for the second: frequency temperature ** (-2 (c // 2) / F), even channels the sine,
        odd channels the cosine of the SAME angle as their even neighbour (position_encoding.py:40-56).  Both functions are
        evaluated for every channel and the parity picks one -- the picked values are the reference's bit for bit (same
        quotient, same function), the layout falls out of a single concatenation instead of interleaving two half-width
        tensors."""
        valid = (~mask).to(torch.float32)
        F = self.num_pos_feats
        channel = torch.arange(F, device=x.device)
        period = self.temperature ** (2 * torch.div(channel, 2, rounding_mode="floor").to(torch.float32) / F)
        take_sin = (channel % 2 == 0)
        halves = []
        for axis in (1, 2):                                        # y (rows), then x (columns)
            count = torch.cumsum(valid, dim=axis)
            if self.normalize:
                last = count.narrow(axis, count.shape[axis] - 1, 1)
                count = count / (last + 1e-6) * self.scale
            angle = count.unsqueeze(-1) / period                   # [B, H, W, F]
            halves.append(torch.where(take_sin, angle.sin(), angle.cos()))
        return torch.cat(halves, dim=3).permute(0, 3, 1, 2)


def build_position_encoding(cfg):
    if cfg["position_embedding"] in ("v2", "sine"):
        return PositionEmbeddingSine(cfg["hidden_dim"] // 2, normalize=True)
    raise ValueError("not supported %s (only the sine encoding is on the shipped path)" % cfg["position_embedding"])
