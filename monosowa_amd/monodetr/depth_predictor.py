"""Depth predictor head: fuses three pyramid levels at stride 16, classifies 80(+1) LID depth
bins per pixel, and builds the depth-aware key/value map of the decoder.

Reference: lib/models/monodetr/depth_predictor/depth_predictor.py:7-104 and
depth_predictor/transformer.py:36-65 (one post-norm MHA encoder layer over the 24x80 = 1920
stride-16 tokens).  Parameter names match the reference modules (``downsample``, ``proj``,
``upsample``, ``depth_head``, ``depth_classifier``, ``depth_encoder.layers.0.*``,
``depth_pos_embed``, ``depth_bin_values``).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..flash_attn import mha_forward, mha_supported
from ..pointwise import conv_channel_bias, conv_group_norm, depth_expectation, dropout_add_layernorm, relu_dropout
from ..token_linear import token_linear


class DepthEncoderLayer(nn.Module):
    """Post-norm transformer encoder layer: q = k = src + pos, v = src (transformer.py:57-65)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, src, src_key_padding_mask, pos):
        qk = src if pos is None else src + pos
        if mha_supported(self.self_attn, qk, qk, src, src_key_padding_mask):
            attn = mha_forward(self.self_attn, qk, qk, src, key_padding_mask=src_key_padding_mask)          # HIP fp32 attention core
        else:
            attn = self.self_attn(qk, qk, value=src, key_padding_mask=src_key_padding_mask, need_weights=False)[0]
        src = dropout_add_layernorm(src, attn, self.norm1, self.dropout1)
        ff = token_linear(relu_dropout(token_linear(src, self.linear1), self.dropout), self.linear2)
        return dropout_add_layernorm(src, ff, self.norm2, self.dropout2)


class DepthEncoder(nn.Module):
    def __init__(self, layer, num_layers):
        super().__init__()
        assert num_layers == 1
        self.layers = nn.ModuleList([layer])
        self.num_layers = num_layers

    def forward(self, src, src_key_padding_mask, pos):
        for layer in self.layers:
            src = layer(src, src_key_padding_mask, pos)
        return src


def lid_bin_values(num_bins, depth_min, depth_max):
    """Centres of the linear-increasing-discretisation bins plus depth_max (depth_predictor.py:21-25)."""
    bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
    idx = torch.linspace(0, num_bins - 1, num_bins)
    values = (idx + 0.5).pow(2) * bin_size / 2 - bin_size / 8 + depth_min
    return torch.cat([values, torch.tensor([depth_max])], dim=0)


FUSED_EXPECTATION = True     # weighted depth as one HIP kernel each way (csrc/ddn_loss.hip)


class DepthPredictor(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        num_bins = int(model_cfg["num_depth_bins"])
        depth_min, depth_max = float(model_cfg["depth_min"]), float(model_cfg["depth_max"])
        self.depth_max = depth_max
        self.depth_bin_values = nn.Parameter(lid_bin_values(num_bins, depth_min, depth_max), requires_grad=False)

        d = model_cfg["hidden_dim"]
        self.downsample = nn.Sequential(nn.Conv2d(d, d, kernel_size=(3, 3), stride=(2, 2), padding=1), nn.GroupNorm(32, d))
        self.proj = nn.Sequential(nn.Conv2d(d, d, kernel_size=(1, 1)), nn.GroupNorm(32, d))
        self.upsample = nn.Sequential(nn.Conv2d(d, d, kernel_size=(1, 1)), nn.GroupNorm(32, d))
        self.depth_head = nn.Sequential(
            nn.Conv2d(d, d, kernel_size=(3, 3), padding=1), nn.GroupNorm(32, num_channels=d), nn.ReLU(),
            nn.Conv2d(d, d, kernel_size=(3, 3), padding=1), nn.GroupNorm(32, num_channels=d), nn.ReLU())
        self.depth_classifier = nn.Conv2d(d, num_bins + 1, kernel_size=(1, 1))
        self.depth_encoder = DepthEncoder(DepthEncoderLayer(d, nhead=8, dim_feedforward=256, dropout=0.1), 1)
        self.depth_pos_embed = nn.Embedding(int(self.depth_max) + 1, 256)

    def forward(self, feature, mask, pos, all_valid=False):
        """feature: 4 projected levels [B,256,H_l,W_l]; mask/pos: of the stride-16 level.
        -> depth_logits [B,bins+1,H,W], depth_embed [B,256,H,W], weighted_depth [B,H,W],
           depth_pos_embed_ip [B,256,H,W]."""
        assert len(feature) == 4
        # Conv2d + GroupNorm(32, d) (+ ReLU) blocks: the norms run through the channels-last HIP kernels
        src_16 = conv_group_norm(feature[1], self.proj[0], self.proj[1])
        src_32 = conv_group_norm(F.interpolate(feature[2], size=src_16.shape[-2:], mode="bilinear"), self.upsample[0], self.upsample[1])
        src_8 = conv_group_norm(feature[0], self.downsample[0], self.downsample[1])
        src = (src_8 + src_16 + src_32) / 3
        src = conv_group_norm(src, self.depth_head[0], self.depth_head[1], relu=True)
        src = conv_group_norm(src, self.depth_head[3], self.depth_head[4], relu=True)
        depth_logits = conv_channel_bias(self.depth_classifier, src)          # (bias gradient: 308 -> 15 us, see pointwise._ChannelBias)

        if FUSED_EXPECTATION:
            weighted_depth = depth_expectation(depth_logits, self.depth_bin_values)      # softmax over the bins x bin centres, summed
        else:
            weighted_depth = (F.softmax(depth_logits, dim=1) * self.depth_bin_values.reshape(1, -1, 1, 1)).sum(dim=1)

        B, C, H, W = src.shape
        tokens = src.flatten(2).permute(2, 0, 1)
        depth_embed = self.depth_encoder(tokens, None if all_valid else mask.flatten(1), pos.flatten(2).permute(2, 0, 1))
        depth_embed = depth_embed.permute(1, 2, 0).reshape(B, C, H, W)
        depth_pos_embed_ip = self.interpolate_depth_embed(weighted_depth)
        return depth_logits, depth_embed + depth_pos_embed_ip, weighted_depth, depth_pos_embed_ip

    def interpolate_depth_embed(self, depth):
        depth = depth.clamp(min=0, max=self.depth_max)
        return self.interpolate_1d(depth, self.depth_pos_embed).permute(0, 3, 1, 2)

    def interpolate_1d(self, coord, embed):
        """Linear interpolation in a learned 1-D table; the floor index is integer bookkeeping
        (depth_predictor.py:99-104: ``embed(floor) * (1 - delta) + embed(ceil) * delta``).

        Evaluated as ``W @ embed.weight`` with W [pixels, table rows] holding the two interpolation weights of
        each pixel: the table has 61 rows, so both the lookup and its gradient (30,720 pixels scattered into 61
        rows; the embedding backward sorts them, 0.7 ms each) are small dense products."""
        floor_coord = coord.floor()
        delta = coord - floor_coord
        floor_idx = floor_coord.long()
        ceil_idx = (floor_idx + 1).clamp(max=embed.num_embeddings - 1)
        n = coord.numel()
        w = torch.zeros((n, embed.num_embeddings), dtype=coord.dtype, device=coord.device)
        w = w.scatter(1, floor_idx.reshape(n, 1), (1 - delta).reshape(n, 1))
        w = w.scatter_add(1, ceil_idx.reshape(n, 1), delta.reshape(n, 1))
        return (w @ embed.weight).view(*coord.shape, embed.embedding_dim)
