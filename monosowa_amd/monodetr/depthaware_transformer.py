"""Depth-aware transformer: 3 visual encoder layers (MSDA self-attention + FFN) and 3 depth-aware
decoder layers (depth cross-attention -> group-wise self-attention -> MSDA cross-attention -> FFN)
with iterative box refinement.

Reference: lib/models/monodetr/depthaware_transformer.py (DepthAwareTransformer :68-312,
VisualEncoderLayer :315-354, VisualEncoder :357-384, DepthAwareDecoderLayer :387-515,
DepthAwareDecoder :518-626, build :644-660).  Only the branch the shipped configs run is restated
(two_stage / use_dab / two_stage_dino are all False in configs/monodetr.yaml:69-74 and in the
released checkpoint's yaml); parameter names -- including the reference's unused ``query_scale``,
``ref_point_head`` and ``sa_v_proj`` -- are kept so that reference state dicts load unchanged.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from .misc import inverse_sigmoid
from .. import MultiScaleDeformableAttention as _MSDA
from ..ms_deform_attn import MSDeformAttn
from .. import encoder_block
from ..flash_attn import mha_forward, mha_supported
from ..pointwise import dropout_add_layernorm, refine_reference, refine_reference_supported, relu_dropout
from ..token_linear import linear as fast_linear, token_linear


LEVEL_EMBED_IN_BLOCK = True   # level_embed gradient from per-level sums inside the encoder blocks (constant pos tensor)
ENCODER_BLOCKS = True     # encoder layers as two autograd nodes whose GEMMs accumulate shared gradients in place
MERGE_HEADS = False       # bbox + dim heads inside the decoder loop as one GEMM: measured neutral (tools/ab_step.py), off
MERGE_SA_PROJ = True      # decoder self-attention: content + positional projections as one GEMM (same input)
SELF_ATTN_HIP = True      # decoder self-attention (50 queries per group) through the HIP attention core (-0.27 ms/step, tools/ab_step.py)


class MLP(nn.Module):
    """Linear-ReLU stack, ReLU on all but the last layer."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = token_linear(x, layer)
            if i < self.num_layers - 1:
                x = F.relu(x)
        return x

    def forward_from_first(self, h):
        """The rest of the stack given ``h = layers[0](x)`` (see ``merged_first_layers``)."""
        for i, layer in enumerate(self.layers[1:], start=1):
            h = token_linear(F.relu(h), layer)
        return h


def merged_first_layers(x, heads):
    """``[head(x) for head in heads]`` for heads (``MLP`` or ``nn.Linear``) that all read the same ``x``: their first
    linear layers run as ONE GEMM over the concatenated weights (one forward launch, one set of backward kernels and one
    gradient for ``x`` instead of one per head), the rest of every MLP continues from its column block."""
    firsts = [h.layers[0] if isinstance(h, MLP) else h for h in heads]
    # widest blocks first (their columns stay 16-byte aligned), and the total padded to a multiple of 4 columns with zero rows:
    # the bias gradient of an odd-width GEMM ([8800, 515].sum(0)) falls off ATen's vectorised reduction (95 us instead of 17)
    order = sorted(range(len(firsts)), key=lambda i: -firsts[i].out_features)
    widths = [firsts[i].out_features for i in order]
    pad = -sum(widths) % 4
    ws, bs = [firsts[i].weight for i in order], [firsts[i].bias for i in order]
    if pad:
        ws.append(ws[0].new_zeros(pad, ws[0].shape[1]))
        bs.append(bs[0].new_zeros(pad))
    parts = fast_linear(x, torch.cat(ws), torch.cat(bs)).split(widths + ([pad] if pad else []), dim=-1)
    by_head = {i: parts[k] for k, i in enumerate(order)}
    return [h.forward_from_first(by_head[i]) if isinstance(h, MLP) else by_head[i] for i, h in enumerate(heads)]


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _add_pos(x, pos):
    return x if pos is None else x + pos


# ----------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------
class VisualEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        assert activation == "relu"
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        if ENCODER_BLOCKS and encoder_block.supported(self, src, pos, reference_points, spatial_shapes, padding_mask):
            out = encoder_block.encoder_layer(self, src, pos, reference_points, spatial_shapes, level_start_index)
            if out is not None:
                return out
        attn = self.self_attn(_add_pos(src, pos), reference_points, src, spatial_shapes, level_start_index, padding_mask)
        src = dropout_add_layernorm(src, attn, self.norm1, self.dropout1)
        ff = token_linear(relu_dropout(token_linear(src, self.linear1), self.dropout2), self.linear2)
        return dropout_add_layernorm(src, ff, self.norm2, self.dropout3)


class VisualEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel-centre grid of every level, normalised by the valid extent, replicated for every
        target level: [B, S, L, 2] (depthaware_transformer.py:363-376)."""
        refs = []
        for lvl, (H_, W_) in enumerate(spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes):
            ys = torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device)
            ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            refs.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(refs, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                spatial_shapes_list=None, unit_ratios=False, level_embed=None):
        """``unit_ratios``: the caller built ``valid_ratios`` as all ones (no padding anywhere), so the reference grid
        depends on the pyramid only and is kept across steps."""
        shapes_for_ref = spatial_shapes_list if spatial_shapes_list is not None else spatial_shapes
        if unit_ratios and spatial_shapes_list is not None:
            key = (tuple(map(tuple, spatial_shapes_list)), valid_ratios.shape[0], src.device)
            cache = self.__dict__.setdefault("_ref_cache", {})
            if key not in cache:
                cache[key] = self.get_reference_points(shapes_for_ref, valid_ratios, device=src.device).detach()
            reference_points = cache[key]
        else:
            reference_points = self.get_reference_points(shapes_for_ref, valid_ratios, device=src.device)
        out = src
        if level_embed is not None and LEVEL_EMBED_IN_BLOCK and ENCODER_BLOCKS and spatial_shapes_list is not None and pos is not None \
                and all(encoder_block.supported(layer, src, pos, reference_points, spatial_shapes, padding_mask) for layer in self.layers):
            # pos = sine encoding + level_embed: hand the layers a constant pos and the parameter itself, so that the
            # level_embed gradient comes from small per-level sums inside the blocks (encoder_block._AttnBlock)
            bounds, start = [], 0
            for h, w in spatial_shapes_list:
                bounds.append((start, start + h * w))
                start += h * w
            pos_const = pos.detach()
            for layer in self.layers:
                nxt = encoder_block.encoder_layer(layer, out, pos_const, reference_points, spatial_shapes, level_start_index,
                                                  level_embed, bounds)
                if nxt is None:
                    break
                out = nxt
            else:
                return out
            out = src                                    # (not reached with the shipped geometry) start over, module by module
        for layer in self.layers:
            out = layer(out, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return out


# ----------------------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------------------
BROADCAST_POS = True         # all-valid masks: one [1, S, C] positional tensor for the whole batch
FUSED_REFINE = True          # the decoder's detached reference refinement as one launch (csrc/head_tail.hip)
MERGE_VALUE_PROJ = True      # the decoder layers' value_proj(memory) as one GEMM


class _MergedValueProj(torch.autograd.Function):
    """memory [B, S, C] x n layers' value_proj -> n views [B, S, C] of ONE [B, S, n*C] projection output (column blocks: token
    stride n*C).  The backward receives the n dense value gradients the operators produce and turns them into d memory
    (accumulating GEMMs), the split-K weight gradients and the column-sum bias gradients -- what n token_linear nodes do,
    minus n - 1 forward launches and the autograd accumulation of d memory."""

    @staticmethod
    def forward(ctx, memory, *params):
        n = len(params) // 2
        weights, biases = params[:n], params[n:]
        C = memory.shape[-1]
        merged = F.linear(memory, torch.cat(weights), torch.cat(biases))                 # [B, S, n*C]
        ctx.save_for_backward(memory, *weights)
        ctx.n = n
        return tuple(merged[..., i * C:(i + 1) * C] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        from ..token_linear import weight_grad
        from ..pointwise import colsum
        memory, *weights = ctx.saved_tensors
        m2 = memory.reshape(-1, memory.shape[-1])
        g2 = [None if g is None else g.reshape(-1, g.shape[-1]) for g in grads]
        gm = None
        if ctx.needs_input_grad[0]:
            for g, w in zip(g2, weights):
                if g is None:
                    continue
                gm = g @ w if gm is None else gm.addmm_(g, w)
            gm = None if gm is None else gm.view_as(memory)
        gw = [None if g is None else weight_grad(g.contiguous(), m2) for g in g2]
        gb = [None if g is None else colsum(g.contiguous()) for g in g2]
        return (gm,) + tuple(gw) + tuple(gb)


def merged_value_proj(memory, attns):
    """-> list of value views (one per MSDeformAttn in ``attns``), or None when the merge does not apply (CPU, other dtypes,
    a single layer): the layers then project for themselves."""
    if len(attns) < 2 or not memory.is_cuda or memory.dtype != torch.float32 or not torch.is_grad_enabled() \
            or any(a.value_proj.bias is None or a.value_proj.weight.shape != attns[0].value_proj.weight.shape for a in attns):
        if len(attns) >= 2 and memory.is_cuda and memory.dtype == torch.float32 and not torch.is_grad_enabled():
            C = memory.shape[-1]
            merged = F.linear(memory, torch.cat([a.value_proj.weight for a in attns]), torch.cat([a.value_proj.bias for a in attns]))
            return [merged[..., i * C:(i + 1) * C] for i in range(len(attns))]
        return None
    return list(_MergedValueProj.apply(memory, *[a.value_proj.weight for a in attns], *[a.value_proj.bias for a in attns]))


class DepthAwareDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4, group_num=1, group_size=50):
        super().__init__()
        assert activation == "relu"
        # visual cross attention
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        # depth cross attention
        self.cross_attn_depth = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout_depth = nn.Dropout(dropout)
        self.norm_depth = nn.LayerNorm(d_model)
        # inter-query self attention
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        # ffn
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

        self.group_num = group_num
        self.group_size = group_size       # the reference hard-codes 50 (depthaware_transformer.py:481-482)
        self.sa_qcontent_proj = nn.Linear(d_model, d_model)
        self.sa_qpos_proj = nn.Linear(d_model, d_model)
        self.sa_kcontent_proj = nn.Linear(d_model, d_model)
        self.sa_kpos_proj = nn.Linear(d_model, d_model)
        self.sa_v_proj = nn.Linear(d_model, d_model)   # present in checkpoints; its output is unused (:471 vs :477)
        self.nhead = n_heads

    def _self_attention(self, tgt, query_pos):
        """q = k = content/pos projections of (tgt + query_pos); v = raw tgt.  In training the
        group_num query groups attend only within their group (folded into the batch)."""
        x = _add_pos(tgt, query_pos)
        v = tgt
        B, Lq, C = tgt.shape
        if self.training:
            G, n = self.group_num, self.group_size
            if Lq != G * n:
                raise NotImplementedError("denoising (extra noise queries) is off in every shipped config; "
                                          "expected %d x %d queries, got %d" % (G, n, Lq))
            fold = lambda t: t.reshape(B * G, n, t.shape[-1]).transpose(0, 1)
        else:
            fold = lambda t: t.transpose(0, 1)
        mha = self.self_attn
        if MERGE_SA_PROJ:
            # content and positional projections see the same input x = tgt + query_pos (depthaware_transformer.py:471-480):
            # x Wc^T + x Wp^T = x (Wc + Wp)^T -- one GEMM per q / k (and one set of backward kernels) instead of two plus an add
            # (-0.65 ms/step; composing further with the attention module's own input projections measured +1.0 ms: the
            # extra 256^3 products and their backward cost more than the two [8800, 256] GEMMs they replace)
            q = fast_linear(x, self.sa_qcontent_proj.weight + self.sa_qpos_proj.weight, self.sa_qcontent_proj.bias + self.sa_qpos_proj.bias)
            k = fast_linear(x, self.sa_kcontent_proj.weight + self.sa_kpos_proj.weight, self.sa_kcontent_proj.bias + self.sa_kpos_proj.bias)
        else:
            q = self.sa_qcontent_proj(x) + self.sa_qpos_proj(x)
            k = self.sa_kcontent_proj(x) + self.sa_kpos_proj(x)
        return self._mha(mha, fold(q), fold(k), fold(v)).transpose(0, 1).reshape(B, Lq, C)

    @staticmethod
    def _mha(mha, q, k, v):
        if SELF_ATTN_HIP and mha_supported(mha, q, k, v):
            return mha_forward(mha, q, k, v)
        return mha(q, k, v, need_weights=False)[0]

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask, depth_pos_embed, mask_depth, value=None):
        # depth cross attention over the stride-16 depth-aware tokens
        tq = tgt.transpose(0, 1)
        if mha_supported(self.cross_attn_depth, tq, depth_pos_embed, depth_pos_embed, mask_depth):
            # (the key padding mask of the depth tokens, depthaware_transformer.py:456-459, goes into the HIP kernels)
            tgt2 = mha_forward(self.cross_attn_depth, tq, depth_pos_embed, depth_pos_embed, key_padding_mask=mask_depth).transpose(0, 1)
        else:
            tgt2 = self.cross_attn_depth(tq, depth_pos_embed, depth_pos_embed,
                                         key_padding_mask=mask_depth, need_weights=False)[0].transpose(0, 1)
        tgt = dropout_add_layernorm(tgt, tgt2, self.norm_depth, self.dropout_depth)
        # self attention
        tgt = dropout_add_layernorm(tgt, self._self_attention(tgt, query_pos), self.norm2, self.dropout2)
        # visual cross attention
        tgt2 = self.cross_attn(_add_pos(tgt, query_pos), reference_points, src, src_spatial_shapes,
                               level_start_index, src_padding_mask, value=value)
        tgt = dropout_add_layernorm(tgt, tgt2, self.norm1, self.dropout1)
        # ffn
        ff = token_linear(relu_dropout(token_linear(tgt, self.linear1), self.dropout3), self.linear2)
        return dropout_add_layernorm(tgt, ff, self.norm3, self.dropout4)


class DepthAwareDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False, d_model=None):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_embed = None     # set by MonoDETR (iterative box refinement)
        self.dim_embed = None
        self.class_embed = None
        # created by the reference in its default branch and never used (:542-544); kept for checkpoints
        self.query_scale = MLP(d_model, d_model, d_model, 2)
        self.ref_point_head = MLP(d_model, d_model, 2, 2)

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None, depth_pos_embed=None, mask_depth=None):
        output = tgt
        inter, inter_refs, inter_dims = [], [], []
        self.bbox_raw = []          # bbox_embed[lid](output) of every layer: MonoDETR.forward needs exactly these again
        # every layer's cross-attention projects the SAME memory with its own value_proj (ms_deform_attn.py:138): one GEMM for
        # all layers, each layer's operator reads its column block in place (value token stride, ABI v7)
        # -- for the layers the fused operator takes: detached reference points (with iterative refinement every layer after the
        # first, :602-613) or 2-d points with a gradient (the first layer's, from the learned query embedding: the operator's
        # autograd node derives d ref from d offsets)
        values = [None] * len(self.layers)
        if MERGE_VALUE_PROJ:
            first = 0 if (not reference_points.requires_grad or reference_points.shape[-1] == 2) \
                else (1 if self.bbox_embed is not None else len(self.layers))
            merged = merged_value_proj(src, [layer.cross_attn for layer in self.layers[first:]])
            if merged is not None:
                values[first:] = merged
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == 6:
                ratios = torch.cat([src_valid_ratios, src_valid_ratios, src_valid_ratios], -1)
                reference_points_input = reference_points[:, :, None] * ratios[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * src_valid_ratios[:, None]
            output = layer(output, query_pos, reference_points_input, src, src_spatial_shapes,
                           src_level_start_index, src_padding_mask, depth_pos_embed, mask_depth,
                           value=values[lid])
            if self.bbox_embed is not None:   # iterative refinement, detached between layers (:602-613)
                if MERGE_HEADS and self.dim_embed is not None:
                    tmp, reference_dims = merged_first_layers(output, [self.bbox_embed[lid], self.dim_embed[lid]])
                else:
                    tmp, reference_dims = self.bbox_embed[lid](output), None
                self.bbox_raw.append(tmp)
                if FUSED_REFINE and refine_reference_supported(tmp, reference_points):
                    reference_points = refine_reference(tmp, reference_points)          # one launch, no graph (detached below anyway)
                elif reference_points.shape[-1] == 6:
                    reference_points = (tmp + inverse_sigmoid(reference_points)).detach().sigmoid()
                else:
                    new_ref = torch.cat([tmp[..., :2] + inverse_sigmoid(reference_points), tmp[..., 2:]], -1)
                    reference_points = new_ref.detach().sigmoid()
            else:
                reference_dims = None
            if reference_dims is None:
                reference_dims = self.dim_embed[lid](output) if self.dim_embed is not None else None
            if self.return_intermediate:
                inter.append(output)
                inter_refs.append(reference_points)
                inter_dims.append(reference_dims)
        if self.return_intermediate:
            # the per-layer tensors themselves, for callers that index the stacks layer by layer (MonoDETR.forward): a
            # `stack[l]` select would zero-fill and copy a full [n_layers, B, Q, C] gradient per layer in the backward
            self.inter_outputs, self.inter_dims = inter, inter_dims
            return torch.stack(inter), torch.stack(inter_refs), torch.stack(inter_dims)
        return output, reference_points


# ----------------------------------------------------------------------------------------------
class DepthAwareTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4, two_stage=False, two_stage_num_proposals=50, group_num=11,
                 use_dab=False, two_stage_dino=False):
        super().__init__()
        if two_stage or use_dab or two_stage_dino:
            raise NotImplementedError("two_stage / use_dab / two_stage_dino are off in every shipped MonoSOWA "
                                      "config and are outside the MI355X hot path")
        self.d_model = d_model
        self.nhead = nhead
        self.two_stage = False
        self.two_stage_num_proposals = two_stage_num_proposals
        self.use_dab = False
        self.two_stage_dino = False
        self.group_num = group_num

        enc_layer = VisualEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points)
        self.encoder = VisualEncoder(enc_layer, num_encoder_layers)
        dec_layer = DepthAwareDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead,
                                           dec_n_points, group_num=group_num, group_size=two_stage_num_proposals)
        self.decoder = DepthAwareDecoder(dec_layer, num_decoder_layers, return_intermediate_dec, d_model)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        constant_(self.reference_points.bias.data, 0.)
        normal_(self.level_embed)

    def _pyramid_tensors(self, shapes, device):
        """int64 [L,2] shapes and [L] level starts on the device, built once per (pyramid, device): they are
        constants of the resolution, a host->device copy per step would also forbid hipGraph capture."""
        cache = self.__dict__.setdefault("_pyramid_cache", {})
        key = (shapes, str(device))
        if key not in cache:
            spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=device)
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            if spatial_shapes.is_cuda:   # the HIP kernels plan their launches from the host copy of the pyramid
                starts = [0]
                for h, w in shapes[:-1]:
                    starts.append(starts[-1] + h * w)
                _MSDA.attach_host_geometry(spatial_shapes, level_start_index, list(shapes), starts)
            cache[key] = (spatial_shapes, level_start_index)
        return cache[key]

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1).float() / H
        valid_w = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([valid_w, valid_h], -1)

    def forward(self, srcs, masks, pos_embeds, query_embed=None, depth_pos_embed=None, depth_pos_embed_ip=None,
                attn_mask=None, all_valid=False):
        """``all_valid``: the caller guarantees every mask is all-False (MonoDETR's backbone builds them so,
        backbone.py:89); padding masks are then not applied at all -- the same result without the masked_fill pass
        over the 10,200 x 256 value tensor in each of the six MSDeformAttn calls, forward and backward."""
        assert query_embed is not None
        src_flat, mask_flat, pos_flat, shapes = [], [], [], []
        for lvl, (src, mask, pos) in enumerate(zip(srcs, masks, pos_embeds)):
            _, _, h, w = src.shape
            shapes.append((h, w))
            src_flat.append(src.flatten(2).transpose(1, 2))
            mask_flat.append(mask.flatten(1))
            if all_valid and BROADCAST_POS:
                pos = pos[:1]       # the sine encoding of an all-valid mask is the same for every image: [1, HW, C], broadcast by its users
            pos_flat.append(pos.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
        src_flat = torch.cat(src_flat, 1)
        mask_flat = torch.cat(mask_flat, 1)
        pos_flat = torch.cat(pos_flat, 1)
        spatial_shapes, level_start_index = self._pyramid_tensors(tuple(shapes), src_flat.device)
        if all_valid:
            mask_flat = None
            valid_ratios = src_flat.new_ones((src_flat.shape[0], len(masks), 2))
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)

        memory = self.encoder(src_flat, spatial_shapes, level_start_index, valid_ratios, pos_flat, mask_flat,
                              spatial_shapes_list=shapes, unit_ratios=all_valid, level_embed=self.level_embed)

        bs, _, c = memory.shape
        query_pos, tgt = torch.split(query_embed, c, dim=1)
        # the learned queries are the same for every image: the reference-point head runs on the [Q, C] embedding once (the
        # reference evaluates it on the expanded [B, Q, C] tensor: same values), and the decoder's first residual stream is a real
        # [B, Q, C] tensor (an expanded view costs every consumer its own copy)
        reference_points = self.reference_points(query_pos).sigmoid().unsqueeze(0).expand(bs, -1, -1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        tgt = tgt.unsqueeze(0).expand(bs, -1, -1).contiguous()
        init_reference_out = reference_points

        depth_tokens = depth_pos_embed.flatten(2).permute(2, 0, 1)
        mask_depth = None if all_valid else masks[1].flatten(1)
        hs, inter_refs, inter_dims = self.decoder(tgt, reference_points, memory, spatial_shapes, level_start_index,
                                                  valid_ratios, query_pos, mask_flat, depth_tokens, mask_depth)
        return hs, init_reference_out, inter_refs, inter_dims, None, None


def build_depthaware_transformer(cfg):
    return DepthAwareTransformer(
        d_model=cfg["hidden_dim"], dropout=cfg["dropout"], activation="relu", nhead=cfg["nheads"],
        dim_feedforward=cfg["dim_feedforward"], num_encoder_layers=cfg["enc_layers"],
        num_decoder_layers=cfg["dec_layers"], return_intermediate_dec=cfg["return_intermediate_dec"],
        num_feature_levels=cfg["num_feature_levels"], dec_n_points=cfg["dec_n_points"],
        enc_n_points=cfg["enc_n_points"], two_stage=cfg["two_stage"], two_stage_num_proposals=cfg["num_queries"],
        use_dab=cfg["use_dab"], two_stage_dino=cfg["two_stage_dino"])
