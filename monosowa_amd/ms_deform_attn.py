"""``MSDeformAttn`` module -- mirror of ops/modules/ms_deform_attn.py:69-162 (same parameter
names, initialisation and forward contract, so reference state dicts load unchanged).

Only the live class is restated; the reference file's ``MSDeformAttn_cross`` and copied
``MultiheadAttention`` (:164-589) are never instantiated by MonoDETR.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import ms_deform_attn_func as _func
from .token_linear import linear as fast_linear, token_linear


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


MASK_IN_KERNEL = True    # fused path: the padding mask is an operator argument instead of a masked_fill pass
MERGED_PROJ = True     # offsets + logits projections as one GEMM in front of the strided fused operator


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-2 head dimension (32 in MonoDETR) takes the fast HIP path")
        self.im2col_step = 64          # ms_deform_attn.py:87
        self.fuse_prologue = True      # MI355X: softmax + sampling locations inside the MSDA kernels when supported
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # ms_deform_attn.py:106-120: zero offset weights, offsets biased to a ring of n_heads
        # directions scaled by the point index; uniform attention; xavier projections.
        constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, value=None):
        """query [N,Lq,C]; reference_points [N,Lq,L,2] or [N,Lq,L,6] (cx,cy,l,r,t,b);
        input_flatten [N,S,C]; input_spatial_shapes [L,2]; input_level_start_index [L];
        input_padding_mask [N,S] (True = padding) -> [N,Lq,C].
        ``value`` (optional, not in the reference): ``self.value_proj(input_flatten)`` computed by the caller, possibly as a
        column block of one GEMM shared by several layers (``merged_value_proj``)."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        # the reference asserts sum(H_l * W_l) == Len_in on the device tensor (ms_deform_attn.py:136), a host
        # synchronisation per call; the host copy of the pyramid answers it for free when the caller attached one
        geom = getattr(input_spatial_shapes, "_msda_host_geometry", None)
        if geom is not None:
            assert sum(geom[0][2 * i] * geom[0][2 * i + 1] for i in range(len(geom[1]))) == Len_in
        else:
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in

        if value is None:
            value = token_linear(input_flatten, self.value_proj)
        H = self.n_heads
        if self.fuse_prologue and MERGED_PROJ and self.n_levels == 4 and self.n_points == 4 and self.d_model == 32 * self.n_heads \
                and query.is_cuda and query.dtype == torch.float32 \
                and (not reference_points.requires_grad or reference_points.shape[-1] == 2) \
                and reference_points.shape[-1] in (2, 6) and geom is not None \
                and _func.MSDeformAttnFunction.__module__ == _func.__name__:
            # sampling_offsets and attention_weights as ONE GEMM; the operator reads (offsets | logits) in place.  The padding
            # mask goes INTO the operator (padded tokens read as zero rows, their gradient rows come out zero: ABI v7) instead of
            # a masked_fill pass over the value tensor, and `value` may be a column block of a wider projection.
            w = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight])
            b = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias])
            proj = fast_linear(query, w, b)
            v4 = value.unflatten(-1, (H, self.d_model // H))                       # a view, also of a column block
            if not (v4.stride(3) == 1 and v4.stride(2) == v4.shape[3] and v4.stride(1) % 4 == 0 and v4.data_ptr() % 16 == 0
                    and (N == 1 or v4.stride(0) == Len_in * v4.stride(1))):
                v4 = v4.contiguous()
            mask = input_padding_mask if (input_padding_mask is not None and MASK_IN_KERNEL) else None
            if input_padding_mask is not None and mask is None:
                v4 = v4.masked_fill(input_padding_mask[..., None, None], float(0))
            output = _func.MSDeformAttnFusedMergedFunction.apply(v4, input_spatial_shapes, input_level_start_index,
                                                                 proj, reference_points.contiguous(), mask)
            return token_linear(output, self.output_proj)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.reshape(N, Len_in, self.n_heads, self.d_model // self.n_heads).contiguous()     # (a caller's column block: copied here)
        sampling_offsets = token_linear(query, self.sampling_offsets).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = token_linear(query, self.attention_weights).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)
        if self.fuse_prologue and _func.MSDA.fused_supported(value, input_spatial_shapes, sampling_offsets, reference_points) \
                and _func.MSDeformAttnFunction.__module__ == _func.__name__:
            # softmax + location arithmetic evaluated inside the kernels (no loc / attn_w tensors in HBM)
            output = _func.MSDeformAttnFusedFunction.apply(
                value.contiguous(), input_spatial_shapes, input_level_start_index, sampling_offsets.contiguous(),
                attention_weights.contiguous(), reference_points.contiguous())
            return token_linear(output, self.output_proj)
        attention_weights = F.softmax(attention_weights, -1).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 6:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.n_points * (reference_points[:, :, None, :, None, 2::2]
                                                      + reference_points[:, :, None, :, None, 3::2]) * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        output = _func.MSDeformAttnFunction.apply(
            value, input_spatial_shapes, input_level_start_index, sampling_locations,
            attention_weights, self.im2col_step)
        return token_linear(output, self.output_proj)
