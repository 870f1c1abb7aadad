"""The two halves of a visual-encoder layer as single autograd nodes (depthaware_transformer.py:339-354 of the reference:
deformable self-attention + post-norm residual, FFN + post-norm residual).

Same kernels and GEMMs as the module-by-module path; what changes is the backward's bookkeeping.  Inside one node the
gradient of a tensor with several consumers (``src`` feeds the residual, ``value_proj`` and -- through ``src + pos`` --
the offset / attention-weight projections; ``src1`` feeds the residual and ``linear1``) is built by GEMMs that accumulate
into the LayerNorm backward's output (``addmm_``, beta = 1) instead of separate gradients that autograd then adds with
[B*S, 256] elementwise passes: 3 of the 4 accumulation passes per layer (75 us each at 163k tokens) disappear, and so do
eight autograd nodes per layer.
"""
import torch
import torch.nn.functional as F

from . import MultiScaleDeformableAttention as MSDA
from .pointwise import colsum, colsum_levels, ln_backward, ln_forward, relu_dropout_backward, relu_dropout_backward_colsum, relu_dropout_forward
from .token_linear import weight_grad


MERGED_PROJ = True      # sampling_offsets and attention_weights as one 384-wide GEMM read in place by the strided operator


FUSED_BIAS_SUMS = True     # bias gradients of output_proj / linear2 / linear1 from the LayerNorm and ReLU backward passes themselves

class _AttnBlock(torch.autograd.Function):
    """LayerNorm(src + dropout(output_proj(MSDA(value_proj(src), offsets(q), logits(q), ref))))"""

    @staticmethod
    def forward(ctx, src, q, ref, shapes, lsi, wv, bv, wo, bo, wa, ba, wp, bp, gamma, beta, p, eps, M, L, P,
                level_embed=None, bounds=None):
        """``level_embed`` / ``bounds``: when given, ``q`` is a constant pos tensor (sine encoding + level_embed values,
        detached) and the node forms q = src + pos itself; the backward then returns the level_embed gradient -- the
        per-level sums of d q -- from small segmented column sums instead of a [B, S, 256] pos gradient that autograd
        would accumulate over the layers and reduce per level with 0.5 TB/s kernels."""
        N, S, C = src.shape
        D = C // M
        ctx.levels = bounds if level_embed is not None else None
        if level_embed is not None:
            q = src + q
        v = F.linear(src, wv, bv).view(N, S, M, D)
        ctx.merged = MERGED_PROJ and L == 4 and P == 4 and D == 32
        ctx.saved_prologue = False
        if ctx.merged:
            # one GEMM for both projections of q; the operator reads (offsets | logits) in place through row strides
            w_ol, b_ol = torch.cat([wo, wa]), torch.cat([bo, ba])
            proj = F.linear(q, w_ol, b_ol)                                        # [N, S, M*48]
            from .ms_deform_attn_func import SAVE_PROLOGUE
            ctx.saved_prologue = SAVE_PROLOGUE and MSDA.fused_save_supported(v, shapes, lsi, S, ref.shape[-1])
            if ctx.saved_prologue:
                # keep the sampling locations / attention weights the kernel evaluated instead of the raw projection
                # (same bytes): the backward's two kernels then skip softmax + location arithmetic (ABI v6)
                a, off, logit = MSDA.ms_deform_attn_fused_forward_merged_save(v, shapes, lsi, proj, ref)
                ctx.msda_plan = MSDA.plan_saved_backward(v, shapes, lsi, off)      # side stream: under the GEMMs that follow
                proj = None
            else:
                a = MSDA.ms_deform_attn_fused_forward_merged(v, shapes, lsi, proj, ref)
                off = logit = None
        else:
            off = F.linear(q, wo, bo).view(N, S, M, L, P, 2)
            logit = F.linear(q, wa, ba).view(N, S, M, L * P)
            a = MSDA.ms_deform_attn_fused_forward(v, shapes, lsi, off, logit, ref)
            proj = w_ol = None
        z = F.linear(a, wp, bp)
        y, s, mean, rstd, seed = ln_forward(src, z, gamma, beta, p, eps)
        ctx.save_for_backward(src, q, ref, shapes, lsi, v, off, logit, proj, w_ol, a, s, mean, rstd, wv, wo, wa, wp, gamma)
        ctx.p, ctx.seed = p, seed
        ctx.host_geom = MSDA.host_geometry(shapes, lsi)
        return y

    @staticmethod
    def backward(ctx, gy):
        src, q, ref, shapes, lsi, v, off, logit, proj, w_ol, a, s, mean, rstd, wv, wo, wa, wp, gamma = ctx.saved_tensors
        C = src.shape[-1]
        sh, ls = ctx.host_geom                                 # the saved pyramid tensors may come back as new objects
        MSDA.attach_host_geometry(shapes, lsi, [(int(sh[2 * i]), int(sh[2 * i + 1])) for i in range(len(ls))], [int(x) for x in ls])
        gx, gz, ggamma, gbeta, gbp = ln_backward(gy, s, mean, rstd, gamma, ctx.p, ctx.seed, with_gz_sum=True)   # gbp: d output_proj.bias
        if not FUSED_BIAS_SUMS:
            gbp = colsum(gz.view(-1, C))
        gz2 = gz.view(-1, C)
        ga = (gz2 @ wp).view_as(a)
        gwp = weight_grad(gz2, a.view(-1, C))
        src2, q2 = src.reshape(-1, C), q.reshape(-1, C)
        n_off = wo.shape[0]
        if ctx.merged:
            if ctx.saved_prologue:          # `off` / `logit` hold the saved sampling locations / attention weights
                gv, gproj = MSDA.ms_deform_attn_fused_backward_merged_saved(v, shapes, lsi, off, logit, ref, ga.contiguous(),
                                                                            plan=getattr(ctx, "msda_plan", None))
            else:
                gv, gproj = MSDA.ms_deform_attn_fused_backward_merged(v, shapes, lsi, proj, ref, ga.contiguous())
            gp2 = gproj.view(-1, gproj.shape[-1])
            # per-level column sums of d proj serve twice: their total is the bias gradient, times W they are d level_embed
            level_sums, level_total = colsum_levels(gproj, ctx.levels, with_total=True) if ctx.levels is not None else (None, None)
            gw_ol = weight_grad(gp2, q2)
            gb_ol = level_total if level_sums is not None else colsum(gp2)
            gwo, gwa, gbo, gba = gw_ol[:n_off], gw_ol[n_off:], gb_ol[:n_off], gb_ol[n_off:]
        else:
            gv, goff, glogit = MSDA.ms_deform_attn_fused_backward(v, shapes, lsi, off, logit, ref, ga.contiguous())
            goff2, glogit2 = goff.view(-1, n_off), glogit.view(-1, wa.shape[0])
            gwo, gbo, gwa, gba = weight_grad(goff2, q2), colsum(goff2), weight_grad(glogit2, q2), colsum(glogit2)
        gv2 = gv.view(-1, C)
        gx2 = gx.view(-1, C)
        gx2.addmm_(gv2, wv)                                  # d src: residual + value path, no separate add pass
        g_level = gq = None
        if ctx.levels is not None:
            # q = src + pos with constant pos: d q lands in d src directly; d level_embed[l] = (sum of d offsets over the
            # level's tokens) @ Wo + (sum of d logits) @ Wa
            if ctx.merged:
                gx2.addmm_(gp2, w_ol)
                g_level = level_sums @ w_ol
            else:
                gx2.addmm_(goff2, wo)
                gx2.addmm_(glogit2, wa)
                g_level = colsum_levels(goff.view(goff.shape[0], goff.shape[1], -1), ctx.levels) @ wo \
                    + colsum_levels(glogit.view(glogit.shape[0], glogit.shape[1], -1), ctx.levels) @ wa
        else:
            if ctx.merged:
                gq = (gp2 @ w_ol).view_as(q)
            else:
                gq = (goff2 @ wo)
                gq.addmm_(glogit2, wa)                       # d q: offsets + attention-weight paths
                gq = gq.view_as(q)
        return (gx, gq, None, None, None, weight_grad(gv2, src2), colsum(gv2), gwo, gbo, gwa, gba, gwp, gbp, ggamma, gbeta,
                None, None, None, None, None, g_level, None)


class _FFNBlock(torch.autograd.Function):
    """LayerNorm(x + dropout(linear2(dropout(relu(linear1(x))))))"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, p_hidden, p_res, eps):
        h = F.linear(x, w1, b1)
        hd = relu_dropout_forward(h, p_hidden) if p_hidden > 0 else torch.relu_(h)
        f = F.linear(hd, w2, b2)
        y, s, mean, rstd, seed = ln_forward(x, f, gamma, beta, p_res, eps)
        ctx.save_for_backward(x, hd, s, mean, rstd, w1, w2, gamma)
        ctx.p_hidden, ctx.p_res, ctx.seed = p_hidden, p_res, seed
        return y

    @staticmethod
    def backward(ctx, gy):
        x, hd, s, mean, rstd, w1, w2, gamma = ctx.saved_tensors
        C, Hd = x.shape[-1], hd.shape[-1]
        gx, gf, ggamma, gbeta, gb2 = ln_backward(gy, s, mean, rstd, gamma, ctx.p_res, ctx.seed, with_gz_sum=True)    # gb2: d linear2.bias
        if not FUSED_BIAS_SUMS:
            gb2 = colsum(gf.view(-1, C))
        gf2 = gf.view(-1, C)
        ghd = (gf2 @ w2).view_as(hd)
        gw2 = weight_grad(gf2, hd.view(-1, Hd))
        if ctx.p_hidden > 0 and Hd == 256 and FUSED_BIAS_SUMS:
            gh, gb1 = relu_dropout_backward_colsum(ghd, hd, ctx.p_hidden)        # d linear1.bias from the same pass
        else:
            gh = relu_dropout_backward(ghd, hd, ctx.p_hidden) if ctx.p_hidden > 0 else ghd * (hd > 0)
            gb1 = colsum(gh.view(-1, Hd))
        gh2 = gh.view(-1, Hd)
        gx.view(-1, C).addmm_(gh2, w1)                       # d x: residual + FFN path
        return gx, weight_grad(gh2, x.reshape(-1, C)), gb1, gw2, gb2, ggamma, gbeta, None, None, None


def supported(layer, src, pos, reference_points, spatial_shapes, padding_mask):
    attn = layer.self_attn
    return (src.is_cuda and src.dtype == torch.float32 and torch.is_grad_enabled() and layer.training and padding_mask is None
            and src.shape[-1] == 256 and attn.d_model == 256 and getattr(attn, "fuse_prologue", False)
            and src.shape[0] * src.shape[1] >= 32768 and layer.linear1.out_features % 4 == 0 and layer.linear1.out_features <= 256
            and reference_points.dtype == torch.float32 and not reference_points.requires_grad
            and all(m.elementwise_affine for m in (layer.norm1, layer.norm2)))


def encoder_layer(layer, src, pos, reference_points, spatial_shapes, level_start_index, level_embed=None, bounds=None):
    """``VisualEncoderLayer.forward`` through the two block nodes (caller checked ``supported``).  With ``level_embed``
    (the parameter) and ``bounds`` (token range of each level), ``pos`` must be the DETACHED sum of the sine encoding and
    the level embedding: its gradient is then produced inside the node (see ``_AttnBlock.forward``)."""
    attn = layer.self_attn
    if level_embed is not None:
        q = pos
    else:
        q = src if pos is None else src + pos
    M, L, P = attn.n_heads, attn.n_levels, attn.n_points
    v_probe = src.new_empty((src.shape[0], src.shape[1], M, 256 // M))
    off_probe = src.new_empty((src.shape[0], src.shape[1], M, L, P, 2))
    if not MSDA.fused_supported(v_probe, spatial_shapes, off_probe, reference_points):
        return None
    src1 = _AttnBlock.apply(src, q, reference_points.contiguous(), spatial_shapes, level_start_index,
                            attn.value_proj.weight, attn.value_proj.bias, attn.sampling_offsets.weight, attn.sampling_offsets.bias,
                            attn.attention_weights.weight, attn.attention_weights.bias, attn.output_proj.weight, attn.output_proj.bias,
                            layer.norm1.weight, layer.norm1.bias, layer.dropout1.p, layer.norm1.eps, M, L, P, level_embed, bounds)
    return _FFNBlock.apply(src1, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                           layer.norm2.weight, layer.norm2.bias, layer.dropout2.p, layer.dropout3.p, layer.norm2.eps)
