"""Autograd glue for MSDA -- mirror of the reference's ``MSDeformAttnFunction``
(ops/functions/ms_deform_attn_func.py:21-38): same ``apply`` signature, saves the same five
tensors, returns ``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.

There is deliberately no pure-PyTorch twin here: the reference's ``ms_deform_attn_core_pytorch``
("for debug and test only", :41-61) lives in ``oracle/`` as the checker.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import MultiScaleDeformableAttention as MSDA


SAVE_PROLOGUE = True     # training, Lq == S: the fused forward saves locations / weights for the backward (ABI v6)


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        # host copy of the pyramid for the backward's launch plan (attached by the transformer that
        # built the tensors; otherwise one tiny device->host read here, not in backward)
        ctx.host_geom = MSDA.host_geometry(value_spatial_shapes, value_level_start_index) if value.is_cuda else None
        output = MSDA.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, ctx.im2col_step, host_geom=ctx.host_geom)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attw = ctx.saved_tensors
        grad_value, grad_loc, grad_attw = MSDA.ms_deform_attn_backward(
            value, shapes, lsi, loc, attw, grad_output.contiguous(), ctx.im2col_step, host_geom=ctx.host_geom)
        return grad_value, None, None, grad_loc, grad_attw, None


class MSDeformAttnFusedFunction(Function):
    """value, shapes, level starts, raw sampling offsets, raw attention logits, per-level reference points ->
    output.  The softmax over the L*P logits and the sampling-location formula of the module
    (ops/modules/ms_deform_attn.py:146-155) run inside the HIP kernels; no gradient flows to the reference
    points (callers check ``MSDA.fused_supported``)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, sampling_offsets, attention_logits, reference_points):
        output = MSDA.ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, sampling_offsets,
                                                   attention_logits, reference_points)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits, reference_points)
        ctx.host_geom = MSDA.host_geometry(spatial_shapes, level_start_index)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, offsets, logits, ref = ctx.saved_tensors
        MSDA.attach_host_geometry(shapes, lsi, *_unpack_geom(ctx.host_geom))
        gv, goff, glog = MSDA.ms_deform_attn_fused_backward(value, shapes, lsi, offsets, logits, ref, grad_output.contiguous())
        return gv, None, None, goff, glog, None


class MSDeformAttnFusedMergedFunction(Function):
    """The fused operator on one merged projection output ``proj`` [B, Lq, M*48] = (sampling offsets | attention logits),
    read and differentiated in place through row strides.  ``value`` [B, S, M, D] may be a column block of a wider projection
    (e.g. one of three 256-column blocks of a [B, S, 768] tensor: its gradient comes back dense), ``value_mask`` [B, S] bool
    marks padded tokens whose value rows count as zero -- ``value.masked_fill(mask[..., None], 0)`` of the reference module
    (ops/modules/ms_deform_attn.py:139-140) without a pass of its own (msda_fused_*_view_f32, ABI v7)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, proj, reference_points, value_mask=None):
        ctx.host_geom = MSDA.host_geometry(spatial_shapes, level_start_index)
        needs_grad = value.requires_grad or proj.requires_grad
        ctx.saved_prologue = SAVE_PROLOGUE and needs_grad and not reference_points.requires_grad and MSDA.fused_save_supported(
            value, spatial_shapes, level_start_index, proj.shape[1], reference_points.shape[-1])
        ctx.value_mask = value_mask
        if ctx.saved_prologue:
            # self-attention shape in training: keep the sampling locations / attention weights the kernel evaluated
            # (250 MB per encoder layer at B = 16) so that neither backward kernel re-evaluates softmax + location math
            output, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, spatial_shapes, level_start_index, proj,
                                                                              reference_points, value_mask)
            ctx.msda_plan = MSDA.plan_saved_backward(value, spatial_shapes, level_start_index, loc)
            ctx.save_for_backward(value, spatial_shapes, level_start_index, loc, attw, reference_points)
        else:
            output = MSDA.ms_deform_attn_fused_forward_merged(value, spatial_shapes, level_start_index, proj, reference_points,
                                                              value_mask)
            ctx.save_for_backward(value, spatial_shapes, level_start_index, proj, reference_points)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if ctx.saved_prologue:
            value, shapes, lsi, loc, attw, ref = ctx.saved_tensors
            MSDA.attach_host_geometry(shapes, lsi, *_unpack_geom(ctx.host_geom))
            gv, gproj = MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, grad_output.contiguous(),
                                                                        ctx.value_mask, plan=getattr(ctx, "msda_plan", None))
            return gv, None, None, gproj, None, None
        value, shapes, lsi, proj, ref = ctx.saved_tensors
        MSDA.attach_host_geometry(shapes, lsi, *_unpack_geom(ctx.host_geom))
        gv, gproj = MSDA.ms_deform_attn_fused_backward_merged(value, shapes, lsi, proj, ref, grad_output.contiguous(), ctx.value_mask)
        g_ref = None
        if ctx.needs_input_grad[4]:
            # 2-d reference points with a gradient (the decoder's first layer: points from the learned query embedding):
            # location = ref + offset / (W_l, H_l), so d ref[b, q, l] = sum over heads and points of d location
            #          = sum of d offset * (W_l, H_l)     (ms_deform_attn.py:149-152)
            assert ref.shape[-1] == 2
            B, Lq, M = gproj.shape[0], gproj.shape[1], value.shape[2]
            g_off = gproj[..., :M * 32].reshape(B, Lq, M, 4, 4, 2)
            normalizer = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(g_off.dtype)
            g_ref = g_off.sum((2, 4)) * normalizer
        return gv, None, None, gproj, g_ref, None


def _unpack_geom(geom):
    sh, ls = geom
    L = len(ls)
    return [(int(sh[2 * i]), int(sh[2 * i + 1])) for i in range(L)], [int(x) for x in ls]
