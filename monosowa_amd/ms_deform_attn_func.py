"""Autograd glue for MSDA -- mirror of the reference's ``MSDeformAttnFunction``
(ops/functions/ms_deform_attn_func.py:21-38): same ``apply`` signature, saves the same five
tensors, returns ``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.

There is deliberately no pure-PyTorch twin here: the reference's ``ms_deform_attn_core_pytorch``
("for debug and test only", :41-61) lives in ``oracle/`` as the checker.
"""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import MultiScaleDeformableAttention as MSDA


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
