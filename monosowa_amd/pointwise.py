"""Fused bias (+ residual) + ReLU after a convolution, in place, through the C-ABI pointwise library
(include/monosowa_pointwise.h).  Used by the backbone's frozen-BN convolutions on channels_last tensors."""
import ctypes
import os

import numpy as np
import torch

from . import flops
from ._lib import raw_stream, on_device

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_pointwise.so")
SYMBOLS = ("mono_bias_act_f32", "mono_bias_relu_maxpool_nhwc_f32", "mono_conv1x1_tail_f32", "mono_conv1x1_tail_ds_f32", "mono_conv1x1_head_f32", "mono_relu_grad_f32", "mono_relu_grad2_f32", "mono_relu_grad3_f32", "mono_bias_relu_mask_f32", "mono_relu_grad_mask_f32", "mono_affine_relu_mask_f32", "mono_affine_relu_grad_f32", "mono_dropout_add_layernorm_fwd_f32",
           "mono_dropout_add_layernorm_bwd_f32", "mono_groupnorm_nhwc_fwd_f32", "mono_groupnorm_nhwc_bwd_f32", "mono_groupnorm_blocks", "mono_colsum_f32", "mono_colsum_strided_f32", "mono_reduce_blocks", "mono_adamw_step_f32", "mono_relu_dropout_fwd_f32",
           "mono_relu_dropout_bwd_f32", "mono_matched_losses_fwd_f32", "mono_matched_losses_bwd_f32", "mono_ddn_loss_blocks",
           "mono_ddn_loss_fwd_f32", "mono_ddn_loss_bwd_f32", "mono_depth_expect_fwd_f32", "mono_depth_expect_bwd_f32", "mono_focal_fwd_f32", "mono_focal_bwd_f32", "mono_head_tail_fwd_f32", "mono_head_tail_bwd_f32", "mono_match_cost_f32", "mono_refine_reference_f32", "mono_relu_dropout_bwd_colsum_f32", "mono_sum_slices_f32", "mono_colsum_any_blocks", "mono_colsum_any_f32", "mono_relu_grad_mask3_f32", "mono_lsap_match_flat_f32",
           "mono_linear_wgrad_workspace", "mono_linear_wgrad_f32", "mono_colsum_levels_blocks", "mono_colsum_levels_f32", "mono_relu_grad_scale_f32")
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("MONOSOWA_POINTWISE_LIB", _PATH)        # another BUILD of the same library (A/B measurements)
        if not os.path.exists(path):
            raise RuntimeError("HIP extension %s is missing: run `python -m monosowa_amd.build`" % path)
        lib = ctypes.CDLL(path)
        P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
        lib.mono_bias_act_f32.restype = I
        lib.mono_bias_act_f32.argtypes = [P, P, P, LL, I, I, P]
        lib.mono_bias_relu_maxpool_nhwc_f32.restype = I
        lib.mono_bias_relu_maxpool_nhwc_f32.argtypes = [P, P, P, I, I, I, I, P]
        lib.mono_conv1x1_tail_f32.restype = I
        lib.mono_conv1x1_tail_f32.argtypes = [P, P, P, P, P, P, LL, I, I, P]
        lib.mono_conv1x1_head_f32.restype = I
        lib.mono_conv1x1_head_f32.argtypes = [P, P, P, P, LL, I, I, P]
        lib.mono_conv1x1_tail_ds_f32.restype = I
        lib.mono_conv1x1_tail_ds_f32.argtypes = [P, P, P, P, P, P, P, LL, I, I, P]
        lib.mono_relu_grad_f32.restype = I
        lib.mono_relu_grad_f32.argtypes = [P, P, P, LL, P]
        lib.mono_relu_grad_scale_f32.restype = I
        lib.mono_relu_grad_scale_f32.argtypes = [P, P, P, P, LL, I, P]
        lib.mono_bias_relu_mask_f32.restype = I
        lib.mono_bias_relu_mask_f32.argtypes = [P, P, P, P, LL, I, P]
        lib.mono_relu_grad_mask_f32.restype = I
        lib.mono_relu_grad_mask_f32.argtypes = [P, P, P, P, LL, P]
        lib.mono_affine_relu_mask_f32.restype = I
        lib.mono_affine_relu_mask_f32.argtypes = [P, P, P, P, LL, I, P]
        lib.mono_affine_relu_grad_f32.restype = I
        lib.mono_affine_relu_grad_f32.argtypes = [P, P, P, P, LL, I, P]
        lib.mono_relu_grad2_f32.restype = I
        lib.mono_relu_grad2_f32.argtypes = [P, P, P, P, LL, P]
        lib.mono_relu_grad3_f32.restype = I
        lib.mono_relu_grad3_f32.argtypes = [P, P, P, P, P, LL, P]
        U, F = ctypes.c_ulonglong, ctypes.c_float
        lib.mono_dropout_add_layernorm_fwd_f32.restype = I
        lib.mono_dropout_add_layernorm_fwd_f32.argtypes = [P] * 8 + [LL, I, F, U, F, P]
        lib.mono_dropout_add_layernorm_bwd_f32.restype = I
        lib.mono_dropout_add_layernorm_bwd_f32.argtypes = [P] * 9 + [LL, I, F, U, P]
        lib.mono_reduce_blocks.restype = I
        lib.mono_reduce_blocks.argtypes = [LL]
        lib.mono_groupnorm_nhwc_fwd_f32.restype = I
        lib.mono_groupnorm_nhwc_fwd_f32.argtypes = [P] * 7 + [I, I, I, I, F, I, P]
        lib.mono_groupnorm_nhwc_bwd_f32.restype = I
        lib.mono_groupnorm_nhwc_bwd_f32.argtypes = [P] * 11 + [I, I, I, I, I, P]
        lib.mono_groupnorm_blocks.restype = I
        lib.mono_groupnorm_blocks.argtypes = [I, I]
        lib.mono_relu_dropout_fwd_f32.restype = I
        lib.mono_relu_dropout_fwd_f32.argtypes = [P, P, LL, F, U, P]
        lib.mono_colsum_any_blocks.restype = I
        lib.mono_colsum_any_blocks.argtypes = [LL]
        lib.mono_colsum_any_f32.restype = I
        lib.mono_colsum_any_f32.argtypes = [P, P, P, LL, I, P]
        lib.mono_relu_grad_mask3_f32.restype = I
        lib.mono_relu_grad_mask3_f32.argtypes = [P, P, P, P, P, LL, P]
        lib.mono_sum_slices_f32.restype = I
        lib.mono_sum_slices_f32.argtypes = [P, P, I, LL, P]
        lib.mono_relu_dropout_bwd_colsum_f32.restype = I
        lib.mono_relu_dropout_bwd_colsum_f32.argtypes = [P, P, P, P, P, LL, F, P]
        lib.mono_relu_dropout_bwd_f32.restype = I
        lib.mono_relu_dropout_bwd_f32.argtypes = [P, P, P, LL, F, P]
        lib.mono_matched_losses_fwd_f32.restype = I
        lib.mono_matched_losses_fwd_f32.argtypes = [P] * 12 + [I] * 4 + [P]
        lib.mono_matched_losses_bwd_f32.restype = I
        lib.mono_matched_losses_bwd_f32.argtypes = [P] * 16 + [I] * 4 + [P]
        lib.mono_adamw_step_f32.restype = I
        lib.mono_adamw_step_f32.argtypes = [P, I] + [ctypes.c_double] * 4 + [P]
        lib.mono_colsum_strided_f32.restype = I
        lib.mono_colsum_strided_f32.argtypes = [P, P, P, I, LL, LL, I, P]
        lib.mono_colsum_levels_blocks.restype = I
        lib.mono_colsum_levels_blocks.argtypes = [I, LL, I, P]
        lib.mono_colsum_levels_f32.restype = I
        lib.mono_colsum_levels_f32.argtypes = [P, P, P, I, LL, I, I, P, I, P]
        lib.mono_colsum_f32.restype = I
        lib.mono_colsum_f32.argtypes = [P, P, P, LL, I, P]
        lib.mono_linear_wgrad_workspace.restype = LL
        lib.mono_linear_wgrad_workspace.argtypes = [I, I, I]
        lib.mono_linear_wgrad_f32.restype = I
        lib.mono_linear_wgrad_f32.argtypes = [P, LL, P, LL, P, P, P, I, I, I, P]
        lib.mono_ddn_loss_blocks.restype = I
        lib.mono_ddn_loss_blocks.argtypes = [I, I, I]
        lib.mono_ddn_loss_fwd_f32.restype = I
        lib.mono_ddn_loss_fwd_f32.argtypes = [P] * 5 + [I] * 5 + [LL] * 3 + [F] * 6 + [P]
        lib.mono_ddn_loss_bwd_f32.restype = I
        lib.mono_ddn_loss_bwd_f32.argtypes = [P] * 6 + [I] * 5 + [LL] * 3 + [F] * 6 + [P]
        lib.mono_depth_expect_fwd_f32.restype = I
        lib.mono_depth_expect_fwd_f32.argtypes = [P] * 3 + [I] * 4 + [LL] * 3 + [P]
        lib.mono_depth_expect_bwd_f32.restype = I
        lib.mono_depth_expect_bwd_f32.argtypes = [P] * 5 + [I] * 4 + [LL] * 3 + [P]
        lib.mono_match_cost_f32.restype = I
        lib.mono_match_cost_f32.argtypes = [P] * 6 + [I] * 5 + [F] * 4 + [P]
        lib.mono_focal_fwd_f32.restype = I
        lib.mono_focal_fwd_f32.argtypes = [P] * 5 + [I] * 5 + [F, F, P]
        lib.mono_focal_bwd_f32.restype = I
        lib.mono_focal_bwd_f32.argtypes = [P] * 5 + [I] * 5 + [F, F, P]
        lib.mono_head_tail_fwd_f32.restype = I
        lib.mono_head_tail_fwd_f32.argtypes = [P] * 8 + [I] * 4 + [P, I, P]
        lib.mono_head_tail_bwd_f32.restype = I
        lib.mono_head_tail_bwd_f32.argtypes = [P] * 12 + [I] * 4 + [P, I, P]
        lib.mono_refine_reference_f32.restype = I
        lib.mono_refine_reference_f32.argtypes = [P, P, P, I, I, P]
        lib.mono_lsap_match_flat_f32.restype = I
        lib.mono_lsap_match_flat_f32.argtypes = [P, I, I, I, I, I, P, P, LL, P, P]
        _lib = lib
    return _lib


def _nhwc_ok(t):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.size(1) % 4 == 0 \
        and t.is_contiguous(memory_format=torch.channels_last) and t.data_ptr() % 16 == 0


USE_RELU_MASK = True      # ReLU backward from a byte mask instead of re-reading the activation (tools/ab_step.py switch)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, residual, relu):
        rows = y.numel() // y.size(1)
        res_ptr = residual.data_ptr() if residual is not None else None
        mask = None
        with on_device(y.device):
            st = raw_stream()
            if relu and USE_RELU_MASK and (y.requires_grad or (residual is not None and residual.requires_grad)):
                mask = torch.empty(y.numel() // 4, dtype=torch.uint8, device=y.device)
                code = load().mono_bias_relu_mask_f32(y.data_ptr(), bias.data_ptr(), res_ptr, mask.data_ptr(), rows, y.size(1), st)
            else:
                code = load().mono_bias_act_f32(y.data_ptr(), bias.data_ptr(), res_ptr, rows, y.size(1), int(relu), st)
        if code:
            raise RuntimeError("mono_bias_act_f32 failed with code %d" % code)
        ctx.mark_dirty(y)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.bias_grad = bias.requires_grad
        ctx.masked = mask is not None
        if relu:
            ctx.save_for_backward(mask if mask is not None else y)
        return y

    @staticmethod
    def backward(ctx, grad):
        if ctx.relu:
            (y,) = ctx.saved_tensors                   # the byte mask when ctx.masked
            grad = grad.contiguous(memory_format=torch.channels_last)
            g = torch.empty_like(grad, memory_format=torch.channels_last)
            with on_device(y.device):
                if ctx.masked:
                    code = load().mono_relu_grad_mask_f32(grad.data_ptr(), None, y.data_ptr(), g.data_ptr(), grad.numel(),
                                                          raw_stream())
                else:
                    code = load().mono_relu_grad_f32(grad.data_ptr(), y.data_ptr(), g.data_ptr(), grad.numel(),
                                                     raw_stream())
            if code:
                raise RuntimeError("mono_relu_grad_f32 failed with code %d" % code)
        else:
            g = grad
        gb = g.sum((0, 2, 3)) if ctx.bias_grad else None
        return g, gb, (g if ctx.has_res else None), None


class _AffineRelu(torch.autograd.Function):
    """relu(y * scale[c] + shift[c]) in place (frozen BN + ReLU after a convolution without residual)."""

    @staticmethod
    def forward(ctx, y, scale, shift):
        rows, C = y.numel() // y.size(1), y.size(1)
        mask = torch.empty(y.numel() // 4, dtype=torch.uint8, device=y.device)
        with on_device(y.device):
            code = load().mono_affine_relu_mask_f32(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), mask.data_ptr(), rows, C,
                                                    raw_stream())
        if code:
            raise RuntimeError("mono_affine_relu_mask_f32 failed with code %d" % code)
        ctx.mark_dirty(y)
        ctx.save_for_backward(mask, scale)
        return y

    @staticmethod
    def backward(ctx, grad):
        mask, scale = ctx.saved_tensors
        grad = grad.contiguous(memory_format=torch.channels_last)
        g = torch.empty_like(grad, memory_format=torch.channels_last)
        with on_device(grad.device):
            code = load().mono_affine_relu_grad_f32(grad.data_ptr(), mask.data_ptr(), scale.data_ptr(), g.data_ptr(),
                                                    grad.numel() // grad.size(1), grad.size(1), raw_stream())
        if code:
            raise RuntimeError("mono_affine_relu_grad_f32 failed with code %d" % code)
        return g, None, None


def affine_relu_supported(y, scale):
    return _nhwc_ok(y) and y.requires_grad and torch.is_grad_enabled() and scale.is_cuda and scale.dtype == torch.float32 \
        and scale.data_ptr() % 16 == 0 and not scale.requires_grad


def affine_relu(y, scale, shift):
    return _AffineRelu.apply(y, scale.contiguous(), shift.contiguous())


class _BiasActFork(torch.autograd.Function):
    """relu(y + bias + residual) in place, returned TWICE (two tensor objects on one storage) for the two consumers of
    a ResNet block output; the backward adds their gradients and applies the ReLU mask in one pass."""

    @staticmethod
    def forward(ctx, y, bias, residual, n_out=2):
        rows = y.numel() // y.size(1)
        mask = torch.empty(y.numel() // 4, dtype=torch.uint8, device=y.device) if USE_RELU_MASK else None
        with on_device(y.device):
            st = raw_stream()
            if mask is not None:
                code = load().mono_bias_relu_mask_f32(y.data_ptr(), bias.data_ptr(), residual.data_ptr(), mask.data_ptr(), rows, y.size(1), st)
            else:
                code = load().mono_bias_act_f32(y.data_ptr(), bias.data_ptr(), residual.data_ptr(), rows, y.size(1), 1, st)
        if code:
            raise RuntimeError("mono_bias_act_f32 failed with code %d" % code)
        ctx.mark_dirty(y)
        ctx.save_for_backward(mask if mask is not None else y)
        ctx.masked = mask is not None
        ctx.shape = y.shape
        ctx.bias_grad = bias.requires_grad
        return (y, y.detach()) if n_out == 2 else (y, y.detach(), y.detach())

    @staticmethod
    def backward(ctx, ga, gb, gc=None):
        (y,) = ctx.saved_tensors                       # the byte mask when ctx.masked
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)
        lib = load()
        g = torch.empty(ctx.shape, dtype=torch.float32, device=y.device, memory_format=torch.channels_last)
        n = g.numel()
        given = [cl(t) for t in (ga, gb, gc) if t is not None]
        if len(given) == 3 and not ctx.masked:
            given = [given[0] + given[1], given[2]]          # (only the byte-mask form has a three-gradient kernel)
        with on_device(y.device):
            st = raw_stream()
            if len(given) == 3:
                code = lib.mono_relu_grad_mask3_f32(given[0].data_ptr(), given[1].data_ptr(), given[2].data_ptr(), y.data_ptr(), g.data_ptr(), n, st)
            elif len(given) == 2:
                if ctx.masked:
                    code = lib.mono_relu_grad_mask_f32(given[0].data_ptr(), given[1].data_ptr(), y.data_ptr(), g.data_ptr(), n, st)
                else:
                    code = lib.mono_relu_grad2_f32(given[0].data_ptr(), given[1].data_ptr(), y.data_ptr(), g.data_ptr(), n, st)
            else:
                if ctx.masked:
                    code = lib.mono_relu_grad_mask_f32(given[0].data_ptr(), None, y.data_ptr(), g.data_ptr(), n, st)
                else:
                    code = lib.mono_relu_grad_f32(given[0].data_ptr(), y.data_ptr(), g.data_ptr(), n, st)
        if code:
            raise RuntimeError("mono_relu_grad(2)_f32 failed with code %d" % code)
        return g, (g.sum((0, 2, 3)) if ctx.bias_grad else None), g, None


def relu_grad_from_output(grads, y, scale=None):
    """``(sum of grads) * (y > 0)`` for the 1 - 3 gradients a ReLU output's consumers returned, in ONE pass that reads the mask off
    the output itself (channels-last float32; the epilogue-GEMM bottleneck paths of backbone.py keep no byte mask).
    ``scale`` (one gradient only): a per-channel factor put on the result in the same pass."""
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    given = [cl(t) for t in grads]
    g = torch.empty(y.shape, dtype=torch.float32, device=y.device, memory_format=torch.channels_last)
    n = g.numel()
    lib = load()
    if scale is not None and len(given) != 1:
        raise ValueError("relu_grad_from_output: a scale goes with exactly one gradient")
    with on_device(y.device):
        if scale is not None:
            code = lib.mono_relu_grad_scale_f32(given[0].data_ptr(), y.data_ptr(), scale.data_ptr(), g.data_ptr(), n, y.shape[1], raw_stream())
        elif len(given) == 3:
            code = lib.mono_relu_grad3_f32(given[0].data_ptr(), given[1].data_ptr(), given[2].data_ptr(), y.data_ptr(), g.data_ptr(), n, raw_stream())
        elif len(given) == 2:
            code = lib.mono_relu_grad2_f32(given[0].data_ptr(), given[1].data_ptr(), y.data_ptr(), g.data_ptr(), n, raw_stream())
        else:
            code = lib.mono_relu_grad_f32(given[0].data_ptr(), y.data_ptr(), g.data_ptr(), n, raw_stream())
    if code:
        raise RuntimeError("mono_relu_grad(2)_f32 failed with code %d" % code)
    return g


def bias_relu_maxpool_supported(y, bias):
    """The one-pass frozen stem applies when nothing upstream wants a gradient (conv1 / bn1 frozen, images are data)."""
    return _nhwc_ok(y) and bias.is_cuda and bias.dtype == torch.float32 and bias.data_ptr() % 16 == 0 and bias.numel() == y.size(1) \
        and not (torch.is_grad_enabled() and (y.requires_grad or bias.requires_grad))


def bias_relu_maxpool(y, bias):
    """``F.max_pool2d(relu(y + bias[None, :, None, None]), kernel_size=3, stride=2, padding=1)`` in one pass (channels-last)."""
    N, C, H, W = y.shape
    out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=y.device, memory_format=torch.channels_last)
    with on_device(y.device):
        code = load().mono_bias_relu_maxpool_nhwc_f32(y.data_ptr(), bias.data_ptr(), out.data_ptr(), N, H, W, C, raw_stream())
    if code:
        raise RuntimeError("mono_bias_relu_maxpool_nhwc_f32 failed with code %d" % code)
    return out


def conv1x1_tail_supported(x, w_kn, residual):
    """The one-pass frozen bottleneck tail: float32 channels-last on the GPU, 64 -> 256 channels, nothing upstream wanting a gradient."""
    return _nhwc_ok(x) and _nhwc_ok(residual) and x.size(1) == 64 and residual.size(1) == 256 and tuple(w_kn.shape) == (64, 256) \
        and x.shape[0] == residual.shape[0] and x.shape[2:] == residual.shape[2:] and w_kn.is_contiguous() and w_kn.data_ptr() % 16 == 0 \
        and not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad or w_kn.requires_grad))


def conv1x1_tail(x, b_in, w_kn, b_out, residual):
    """``relu(conv1x1(relu(x + b_in), w) + b_out + residual)`` in one pass; w_kn = the convolution's weight as [in, out]."""
    out = torch.empty_like(residual)
    M = x.numel() // 64
    flops.conv1x1(M, 64, 256)
    with on_device(x.device):
        code = load().mono_conv1x1_tail_f32(x.data_ptr(), b_in.data_ptr(), w_kn.data_ptr(), b_out.data_ptr(), residual.data_ptr(),
                                            out.data_ptr(), M, 64, 256, raw_stream())
    if code:
        raise RuntimeError("mono_conv1x1_tail_f32 failed with code %d" % code)
    return out


def conv1x1_tail_ds_supported(x, w_kn, x0, wd_kn):
    ok_w = lambda w: tuple(w.shape) == (64, 256) and w.is_contiguous() and w.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and w.requires_grad)
    return _nhwc_ok(x) and _nhwc_ok(x0) and x.size(1) == 64 and x0.shape == x.shape and ok_w(w_kn) and ok_w(wd_kn) \
        and not (torch.is_grad_enabled() and (x.requires_grad or x0.requires_grad))


def conv1x1_tail_ds(x, b_in, w_kn, x0, wd_kn, b_out):
    """``relu(conv1x1(relu(x + b_in), w) + conv1x1(x0, wd) + b_out)`` in one pass; weights as [in, out]."""
    N, _, H, W = x.shape
    out = torch.empty((N, 256, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    flops.conv1x1(N * H * W, 128, 256)
    with on_device(x.device):
        code = load().mono_conv1x1_tail_ds_f32(x.data_ptr(), b_in.data_ptr(), w_kn.data_ptr(), x0.data_ptr(), wd_kn.data_ptr(), b_out.data_ptr(),
                                               out.data_ptr(), N * H * W, 64, 256, raw_stream())
    if code:
        raise RuntimeError("mono_conv1x1_tail_ds_f32 failed with code %d" % code)
    return out


def conv1x1_head_supported(x, w_kn):
    return _nhwc_ok(x) and x.size(1) in (64, 256) and tuple(w_kn.shape) == (x.size(1), 64) and w_kn.is_contiguous() \
        and w_kn.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and (x.requires_grad or w_kn.requires_grad))


def conv1x1_head(x, w_kn, b_out):
    """``relu(conv1x1(x, w) + b_out)`` in one pass (64 output channels); w_kn = the convolution's weight as [in, out]."""
    N, K, H, W = x.shape
    out = torch.empty((N, 64, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    flops.conv1x1(N * H * W, K, 64)
    with on_device(x.device):
        code = load().mono_conv1x1_head_f32(x.data_ptr(), w_kn.data_ptr(), b_out.data_ptr(), out.data_ptr(), N * H * W, K, 64, raw_stream())
    if code:
        raise RuntimeError("mono_conv1x1_head_f32 failed with code %d" % code)
    return out


def bias_act_fork(y, bias, residual, n_out=2):
    """``relu(y + bias + residual)`` ``n_out`` (2 or 3) times -- the same values as separate tensor objects, one per consumer
    (see ``_BiasActFork``); plain PyTorch (the same tensor repeated) off the HIP path."""
    if _nhwc_ok(y) and bias.is_cuda and bias.dtype == torch.float32 and bias.data_ptr() % 16 == 0 \
            and _nhwc_ok(residual) and residual.shape == y.shape and torch.is_grad_enabled() and y.requires_grad:
        return _BiasActFork.apply(y, bias.contiguous(), residual, n_out)
    out = bias_act(y, bias, residual, True)
    return (out,) * n_out


def bias_act(y, bias, residual=None, relu=True):
    """``relu(y + bias[None, :, None, None] (+ residual))`` -- in place on ``y`` through the HIP kernel when the
    tensors are float32 channels_last on the GPU, with plain PyTorch ops otherwise."""
    if _nhwc_ok(y) and bias.is_cuda and bias.dtype == torch.float32 and bias.data_ptr() % 16 == 0 \
            and (residual is None or (_nhwc_ok(residual) and residual.shape == y.shape)):
        return _BiasAct.apply(y, bias.contiguous(), residual, relu)
    out = y + bias.view(1, -1, 1, 1)
    if residual is not None:
        out = out + residual
    return torch.relu(out) if relu else out


# ---------------------------------------------------------------------------------------------------------
_seed_counter = [0]


def _next_seed():
    """A fresh 64-bit seed per call, reproducible under torch.manual_seed and distinct across ranks."""
    _seed_counter[0] += 1
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0] * 0xD1B54A32D192ED03 + rank * 0x94D049BB133111EB) & (2 ** 64 - 1)


LN_MIN_ROWS = int(os.environ.get('MONOSOWA_LN_MIN_ROWS', '1'))


def row_dense(t):
    """A tensor whose elements fill its storage without gaps or overlap, with the last dimension contiguous, in SOME order of the
    leading dimensions (e.g. the [L, B, C] view of a [B, L, C] buffer): per-row and elementwise kernels can walk it in memory
    order, and ``torch.empty_like`` reproduces its strides."""
    if t.is_contiguous():
        return True
    if t.dim() < 2 or t.stride(-1) != 1:
        return False
    order = sorted(range(t.dim() - 1), key=lambda d: -t.stride(d))
    return t.permute(*order, t.dim() - 1).is_contiguous()


def same_layout(ref, t):
    """``t`` in the memory layout of the row-dense tensor ``ref`` (a copy only if it is not already)."""
    if t.stride() == ref.stride():
        return t
    return torch.empty_like(ref).copy_(t)


def ln_forward(x, z, weight, bias, p, eps):
    """y = LayerNorm_256(x + dropout_p(z)); returns (y, s, mean, rstd, seed) -- the last four feed ``ln_backward``."""
    # rows are independent: any row-dense layout is walked in memory order (the result keeps it), e.g. the [L, B, C] view of a
    # batch-major buffer -- no copy into [L, B, C] order and back
    if not row_dense(x):
        x = x.contiguous()
    z = same_layout(x, z)
    rows = x.numel() // 256
    y, s = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    seed = _next_seed() if p > 0 else 0
    with on_device(x.device):
        code = load().mono_dropout_add_layernorm_fwd_f32(
            x.data_ptr(), z.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), s.data_ptr(), mean.data_ptr(),
            rstd.data_ptr(), rows, 256, float(p), seed, float(eps), raw_stream())
    if code:
        raise RuntimeError("mono_dropout_add_layernorm_fwd_f32 failed with code %d" % code)
    return y, s, mean, rstd, seed


def ln_backward(gy, s, mean, rstd, weight, p, seed, with_gz_sum=False):
    """-> (gx, gz, gweight, gbias) of ``ln_forward``; with_gz_sum: also gz summed over the rows (the kernel accumulates it
    anyway: it is the bias gradient of the linear layer whose output z is -- a 167 MB pass saved per encoder norm)."""
    gy = same_layout(s, gy)
    gx, gz = torch.empty_like(s), torch.empty_like(s)
    rows = s.numel() // 256
    gw = torch.empty(3, 256, dtype=torch.float32, device=s.device)
    partials = torch.empty(load().mono_reduce_blocks(rows) * 768, dtype=torch.float32, device=s.device)
    with on_device(s.device):
        code = load().mono_dropout_add_layernorm_bwd_f32(
            gy.data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(), weight.data_ptr(), gx.data_ptr(), gz.data_ptr(),
            gw.data_ptr(), partials.data_ptr(), rows, 256, float(p), seed, raw_stream())
    if code:
        raise RuntimeError("mono_dropout_add_layernorm_bwd_f32 failed with code %d" % code)
    if with_gz_sum:
        return gx, gz, gw[0], gw[1], gw[2]
    return gx, gz, gw[0], gw[1]


class _DropoutAddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z, weight, bias, p, eps):
        y, s, mean, rstd, seed = ln_forward(x, z, weight, bias, p, eps)
        ctx.save_for_backward(s, mean, rstd, weight)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, gy):
        s, mean, rstd, weight = ctx.saved_tensors
        return ln_backward(gy, s, mean, rstd, weight, ctx.p, ctx.seed) + (None, None)


def dropout_add_layernorm(x, z, norm, dropout):
    """``norm(x + dropout(z))`` for an ``nn.LayerNorm(256)`` and an ``nn.Dropout``: one HIP kernel forward, one
    backward, on float32 GPU tensors; the PyTorch formulation otherwise."""
    p = dropout.p if dropout.training else 0.0
    # A/B switch (tools/ab_step.py): even for the 8,800-token decoder, where the step is host-bound behind the matcher's
    # sync and a Python autograd node costs more host time than three ATen ops, the fused kernels win by 0.47 ms/step
    if x.numel() < LN_MIN_ROWS * 256 and torch.is_grad_enabled():
        return norm(x + dropout(z))
    if x.is_cuda and x.dtype == torch.float32 and z.dtype == torch.float32 and x.shape == z.shape and x.shape[-1] == 256 \
            and tuple(norm.normalized_shape) == (256,) and norm.elementwise_affine and norm.bias is not None and p < 1.0:
        return _DropoutAddLayerNorm.apply(x, z, norm.weight, norm.bias, p, norm.eps)
    return norm(x + dropout(z))


# ---------------------------------------------------------------------------------------------------------
_ZERO_POOL = {}          # device -> [chunk, used]: float64 zeros handed out in slices, never reused (a new chunk when one is used up)
ZERO_POOL_DOUBLES = 1 << 17   # 1 MiB per chunk: the nine GroupNorms of a train step take 0.65 MB of zeroed accumulators


def zeros_f64(n, device):
    """``n`` float64 zeros on ``device`` (16-byte aligned), cut from a pooled chunk: ONE fill launch per chunk instead of one per
    accumulator (the GroupNorm kernels add into zeroed f64 statistics: 18 tiny fills per step).  A slice is handed out once; the
    chunk lives as long as any of its slices.  Under hipGraph capture every accumulator gets its own captured fill."""
    n_al = (n + 1) & ~1
    # under stream capture the fill has to be a node of the graph (a replay must start from zeros again): no pooling there
    if n_al > ZERO_POOL_DOUBLES // 4 or (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        return torch.zeros(n, dtype=torch.float64, device=device)
    key = (device.type, device.index)
    slot = _ZERO_POOL.get(key)
    if slot is None or slot[1] + n_al > ZERO_POOL_DOUBLES:
        slot = _ZERO_POOL[key] = [torch.zeros(ZERO_POOL_DOUBLES, dtype=torch.float64, device=device), 0]
    out = slot[0][slot[1]:slot[1] + n]
    slot[1] += n_al
    return out


class _GroupNormNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre_bias, weight, bias, eps, relu):
        B, C, H, W = x.shape
        y = torch.empty_like(x)                                       # keeps the channels_last strides
        stats = zeros_f64(B * 32 * 2, x.device)
        mean_rstd = torch.empty(B, 32, 2, dtype=torch.float32, device=x.device)
        with on_device(x.device):
            code = load().mono_groupnorm_nhwc_fwd_f32(x.data_ptr(), pre_bias.data_ptr() if pre_bias is not None else None,
                                                      weight.data_ptr(), bias.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                                      mean_rstd.data_ptr(), B, H * W, C, 32, float(eps), int(relu),
                                                      raw_stream())
        if code:
            raise RuntimeError("mono_groupnorm_nhwc_fwd_f32 failed with code %d" % code)
        ctx.relu = relu
        ctx.save_for_backward(x, pre_bias, y if relu else None, mean_rstd, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, pre_bias, y, mean_rstd, weight = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        part = zeros_f64(B * C * 2, x.device)
        gwb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        lib = load()
        gbias = partials = None
        if pre_bias is not None:
            gbias = torch.empty(C, dtype=torch.float32, device=x.device)
            partials = torch.empty(lib.mono_groupnorm_blocks(B, H * W) * C, dtype=torch.float32, device=x.device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with on_device(x.device):
            code = lib.mono_groupnorm_nhwc_bwd_f32(gy.data_ptr(), x.data_ptr(), ptr(pre_bias), ptr(y if ctx.relu else None),
                                                   mean_rstd.data_ptr(), weight.data_ptr(), gx.data_ptr(), part.data_ptr(),
                                                   ptr(gbias), ptr(partials), gwb.data_ptr(), B, H * W, C, 32, int(ctx.relu),
                                                   raw_stream())
        if code:
            raise RuntimeError("mono_groupnorm_nhwc_bwd_f32 failed with code %d" % code)
        return gx, gbias, gwb[0], gwb[1], None, None      # ggamma / gbeta: two contiguous rows written by the backward's second kernel


def group_norm(x, gn, relu=False, pre_bias=None):
    """``gn(x + pre_bias[None, :, None, None])`` (+ ReLU) for an ``nn.GroupNorm(32, 256)`` on a channels-last float32 GPU
    tensor: two HIP kernels forward, two backward, no layout copies; the PyTorch formulation for anything else.
    ``pre_bias``: the bias of the convolution in front, see ``conv_group_norm``."""
    if _nhwc_ok(x) and x.size(1) == 256 and gn.num_groups == 32 and gn.affine and x.size(0) <= 65535:
        return _GroupNormNHWC.apply(x, pre_bias, gn.weight, gn.bias, gn.eps, relu)
    if pre_bias is not None:
        x = x + pre_bias.view(1, -1, 1, 1)
    y = gn(x)
    return torch.relu(y) if relu else y


def conv_group_norm(x, conv, gn, relu=False):
    """``gn(conv(x))`` (+ ReLU) for ``nn.Sequential(Conv2d, GroupNorm(32, 256))``: the convolution runs without its bias,
    which is added inside the normalisation kernels; its gradient is the per-channel sum of the kernels' grad_input
    (PyTorch: a bias pass after the convolution and a 0.5 TB/s reduction for its gradient)."""
    if conv.bias is not None and x.is_cuda and x.dtype == torch.float32 and conv.out_channels == 256 and gn.num_groups == 32:
        y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if _nhwc_ok(y):
            return group_norm(y, gn, relu, pre_bias=conv.bias)
        return group_norm(y + conv.bias.view(1, -1, 1, 1), gn, relu)
    return group_norm(conv(x), gn, relu)


COLSUM_LEVELS_ONE_LAUNCH = 1     # all levels by one launch pair (mono_colsum_levels_f32); 0: a launch pair per level
_LEVEL_PLANS = {}                # (B, S, bounds) -> (ctypes int array, partial rows)


def colsum_levels(g3, bounds, with_total=False):
    """Per-level column sums of a [B, S, C] tensor (contiguous, C <= 512): rows [a, b) of every batch for each (a, b)
    in ``bounds`` -> [len(bounds), C].  ``with_total``: -> (the same, their sum over the levels [C]) -- from the same launch pair."""
    B, S, C = g3.shape
    L = len(bounds)
    out = torch.empty((L + (1 if with_total else 0), C), dtype=torch.float32, device=g3.device)
    lib = load()
    if COLSUM_LEVELS_ONE_LAUNCH and L <= 8 and C % 4 == 0 and C <= 512:
        key = (B, S, tuple(tuple(ab) for ab in bounds))
        plan = _LEVEL_PLANS.get(key)
        if plan is None:
            arr = (ctypes.c_int * (2 * L))(*[int(v) for ab in bounds for v in ab])
            plan = _LEVEL_PLANS[key] = (arr, lib.mono_colsum_levels_blocks(B, S, L, arr))
        arr, blocks = plan
        if blocks > 0:
            partials = torch.empty(blocks * C, dtype=torch.float32, device=g3.device)
            with on_device(g3.device):
                code = lib.mono_colsum_levels_f32(g3.data_ptr(), out.data_ptr(), partials.data_ptr(), B, S, C, L, arr, int(with_total), raw_stream())
            if code:
                raise RuntimeError("mono_colsum_levels_f32 failed with code %d" % code)
            return (out[:L], out[L]) if with_total else out
    with on_device(g3.device):
        st = raw_stream()
        for i, (a, b) in enumerate(bounds):
            partials = torch.empty(lib.mono_reduce_blocks(B * (b - a)) * C, dtype=torch.float32, device=g3.device)
            code = lib.mono_colsum_strided_f32(g3.data_ptr() + a * C * 4, out[i].data_ptr(), partials.data_ptr(), B, b - a, S * C, C, st)
            if code:
                raise RuntimeError("mono_colsum_strided_f32 failed with code %d" % code)
    if with_total:
        return out[:L], out[:L].sum(0)
    return out


class _ChannelBias(torch.autograd.Function):
    """y + bias[None, :, None, None] in place on a convolution's fresh output; the bias gradient is summed over the rows of the
    channels-last gradient seen as [N H W, C] -- ATen's `sum((0, 2, 3))` of a channels-last [16, 81, 24, 80] tensor (the bias
    gradient inside ConvolutionBackward) takes 308 us."""

    @staticmethod
    def forward(ctx, y, bias):
        ctx.mark_dirty(y)
        return y.add_(bias.view(1, -1, 1, 1))

    @staticmethod
    def backward(ctx, g):
        C = g.shape[1]
        if g.is_contiguous(memory_format=torch.channels_last):
            g2 = g.permute(0, 2, 3, 1).reshape(-1, C)                # a view: [N H W, C] row-major
            # (ATen's reduction of this odd-width matrix -- 81 depth bins -- is as slow as the 4-d form: 308 us; rocBLAS gemv 320)
            gb = colsum(g2)
        else:
            gb = g.sum((0, 2, 3))
        return g, gb


def conv_channel_bias(conv, x):
    """``conv(x)`` for an ``nn.Conv2d`` with a bias: bias-free convolution + the bias as its own node (see ``_ChannelBias``)."""
    if conv.bias is None or not (x.is_cuda and torch.is_grad_enabled() and conv.bias.requires_grad):
        return conv(x)
    y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return _ChannelBias.apply(y, conv.bias)


def sum_slices(t):
    """``t.sum(0)`` of a contiguous float32 GPU stack [n, ...] (the slices of a split-K weight gradient): one coalesced pass; the
    generic reduction over dim 0 reaches 0.24 TB/s on [64, 256, 256]."""
    inner = t[0].numel() if t.dim() > 1 and t.size(0) > 0 else 0
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and inner and inner % 4 == 0 and t.data_ptr() % 16 == 0:
        out = torch.empty(t.shape[1:], dtype=torch.float32, device=t.device)
        with on_device(t.device):
            code = load().mono_sum_slices_f32(t.data_ptr(), out.data_ptr(), t.size(0), inner, raw_stream())
        if code:
            raise RuntimeError("mono_sum_slices_f32 failed with code %d" % code)
        return out
    return t.sum(0)


COLSUM_MAX_C = 512     # wider matrices: the PyTorch reduction is as fast


_WGRAD_WS = {}          # (rows, out, in) -> floats of workspace (0: shape not served); saves a library call per launch


def _wgrad_workspace(R, M, N):
    n = _WGRAD_WS.get((R, M, N))
    if n is None:
        n = _WGRAD_WS[(R, M, N)] = load().mono_linear_wgrad_workspace(R, M, N)
    return n


def linear_wgrad_applies(g2, x2):
    """True when ``linear_wgrad`` serves the pair: f32 GPU matrices with unit column stride, 16-byte aligned rows, out / in features
    multiples of 64, at least 64 rows."""
    if not (g2.is_cuda and g2.dtype == torch.float32 and x2.dtype == torch.float32 and g2.dim() == 2 and x2.dim() == 2):
        return False
    (R, M), (R2, N) = g2.shape, x2.shape
    sg, sx = g2.stride(), x2.stride()
    return R == R2 and sg[1] == 1 and sx[1] == 1 and sg[0] >= M and sx[0] >= N and sg[0] % 4 == 0 and sx[0] % 4 == 0 \
        and g2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and _wgrad_workspace(R, M, N) > 0


def linear_wgrad(g2, x2, with_bias=True, checked=False):
    """(g2^T @ x2, g2.sum(0) or None) for [rows, out] / [rows, in] matrices by `mono_linear_wgrad_f32` (csrc/small_wgrad.hip): the weight and
    bias gradients of a linear over a few thousand tokens, two launches, dY read once.  Raises when the shape is not served
    (``checked``: the caller has asked ``linear_wgrad_applies`` already -- the host side of this call is on the step's critical path)."""
    if not checked and not linear_wgrad_applies(g2, x2):
        raise ValueError("linear_wgrad: unsupported operands %s %s" % (tuple(g2.shape), tuple(x2.shape)))
    R, M = g2.shape
    N = x2.shape[1]
    flops.linear_wgrad(R, M, N)
    ws = torch.empty(_wgrad_workspace(R, M, N), dtype=torch.float32, device=g2.device)
    out = torch.empty(M * N + M, dtype=torch.float32, device=g2.device)
    p = out.data_ptr()
    with on_device(g2.device):
        code = load().mono_linear_wgrad_f32(g2.data_ptr(), g2.stride(0), x2.data_ptr(), x2.stride(0), p, p + 4 * M * N if with_bias else None,
                                            ws.data_ptr(), R, M, N, raw_stream())
    if code:
        raise RuntimeError("mono_linear_wgrad_f32 failed with code %d" % code)
    return out[:M * N].view(M, N), (out[M * N:] if with_bias else None)


def colsum(g2):
    """``g2.sum(0)`` of a [rows, C] matrix (bias gradients); HIP kernel for contiguous float32 GPU input with
    C % 4 == 0 and C <= 512 (wider matrices: the PyTorch reduction is as fast), torch otherwise."""
    rows, C = (g2.shape if g2.dim() == 2 else (0, 0))
    if rows > 0 and C % 4 == 0 and C <= COLSUM_MAX_C and g2.is_cuda and g2.dtype == torch.float32 and g2.is_contiguous() \
            and g2.data_ptr() % 16 == 0:
        lib = load()
        # one allocation: [out | partial rows]  (the host side of this call is on the step's critical path, see token_linear)
        buf = torch.empty((lib.mono_reduce_blocks(rows) + 1) * C, dtype=torch.float32, device=g2.device)
        p0 = buf.data_ptr()
        if g2.device.index == torch.cuda.current_device():
            code = lib.mono_colsum_f32(g2.data_ptr(), p0, p0 + 4 * C, rows, C, raw_stream())
        else:
            with on_device(g2.device):
                code = lib.mono_colsum_f32(g2.data_ptr(), p0, p0 + 4 * C, rows, C, raw_stream())
        if code:
            raise RuntimeError("mono_colsum_f32 failed with code %d" % code)
        return buf[:C]
    if g2.is_cuda and g2.dtype == torch.float32 and g2.dim() == 2 and g2.is_contiguous() and 16 <= g2.size(1) <= 1024 and g2.size(1) % 4 \
            and g2.size(0) > 0:
        lib = load()                                   # odd widths fall off ATen's vectorised reduction: 308 us for [30720, 81], 59 here
        out = torch.empty(g2.size(1), dtype=torch.float32, device=g2.device)
        partials = torch.empty(lib.mono_colsum_any_blocks(g2.size(0)) * g2.size(1), dtype=torch.float32, device=g2.device)
        with on_device(g2.device):
            code = lib.mono_colsum_any_f32(g2.data_ptr(), out.data_ptr(), partials.data_ptr(), g2.size(0), g2.size(1), raw_stream())
        if code:
            raise RuntimeError("mono_colsum_any_f32 failed with code %d" % code)
        return out
    return g2.sum(0)


# ---------------------------------------------------------------------------------------------------------
ADAM_CHUNK = 32768


class FusedAdamWPlan:
    """Chunk table of one parameter group for ``mono_adamw_step_f32``: the split into chunks is fixed, the four
    address columns are refreshed every step (gradients are re-allocated) and shipped with one non-blocking copy."""

    def __init__(self, params, exp_avgs, exp_avg_sqs, weight_decay):
        import numpy as np
        sizes = np.array([p.numel() for p in params], dtype=np.int64)
        per = (sizes + ADAM_CHUNK - 1) // ADAM_CHUNK
        self.tensor = np.repeat(np.arange(len(params)), per)
        first = np.cumsum(per) - per
        self.offset = (np.arange(per.sum()) - np.repeat(first, per)) * ADAM_CHUNK
        self.n = np.minimum(sizes[self.tensor] - self.offset, ADAM_CHUNK).astype(np.int32)
        self.n_chunks = int(per.sum())
        self.device = params[0].device
        ptr = lambda ts: np.array([t.data_ptr() for t in ts], dtype=np.uint64)
        self.fixed = [ptr(params)[self.tensor] + (self.offset * 4).astype(np.uint64),
                      ptr(exp_avgs)[self.tensor] + (self.offset * 4).astype(np.uint64),
                      ptr(exp_avg_sqs)[self.tensor] + (self.offset * 4).astype(np.uint64)]
        self.keys = (ptr(params), ptr(exp_avgs), ptr(exp_avg_sqs))
        nbytes = self.n_chunks * (4 * 8 + 4 + 4)
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.view = self.host.numpy()
        u64 = self.view[:self.n_chunks * 32].view(np.uint64).reshape(4, self.n_chunks)
        u64[0], u64[2], u64[3] = self.fixed
        self.g_col = u64[1]
        self.view[self.n_chunks * 32:self.n_chunks * 36].view(np.int32)[:] = self.n
        self.view[self.n_chunks * 36:].view(np.float32)[:] = weight_decay
        self.np = np

    def matches(self, params, exp_avgs, exp_avg_sqs):
        ptr = lambda ts: self.np.array([t.data_ptr() for t in ts], dtype=self.np.uint64)
        return all(self.np.array_equal(a, ptr(b)) for a, b in zip(self.keys, (params, exp_avgs, exp_avg_sqs)))

    def step(self, grads, beta1, beta2, eps, step_size):
        np = self.np
        g = np.array([t.data_ptr() for t in grads], dtype=np.uint64)
        if getattr(self, "copied", None) is not None:
            self.copied.synchronize()                     # the previous step's table copy has left the pinned buffer
        self.g_col[:] = g[self.tensor] + (self.offset * 4).astype(np.uint64)
        self.dev.copy_(self.host, non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()
        with on_device(self.device):
            code = load().mono_adamw_step_f32(self.dev.data_ptr(), self.n_chunks, beta1, beta2, eps, step_size,
                                              raw_stream())
        if code:
            raise RuntimeError("mono_adamw_step_f32 failed with code %d" % code)


# ---------------------------------------------------------------------------------------------------------
def relu_dropout_forward(h, p):
    if not row_dense(h):
        h = h.contiguous()
    y = torch.empty_like(h)                      # elementwise: memory order, strides kept
    with on_device(h.device):
        code = load().mono_relu_dropout_fwd_f32(h.data_ptr(), y.data_ptr(), h.numel(), float(p), _next_seed(), raw_stream())
    if code:
        raise RuntimeError("mono_relu_dropout_fwd_f32 failed with code %d" % code)
    return y


def relu_dropout_backward_colsum(gy, y, p):
    """-> (grad_h, grad_h summed over the rows) for a [..., 256] activation: one pass for both (the sum is the bias gradient
    of the linear in front)."""
    assert y.shape[-1] == 256
    gy = same_layout(y, gy)
    gh = torch.empty_like(y)
    rows = y.numel() // 256
    lib = load()
    out = torch.empty(256, dtype=torch.float32, device=y.device)
    partials = torch.empty(lib.mono_reduce_blocks(rows) * 256, dtype=torch.float32, device=y.device)
    with on_device(y.device):
        code = lib.mono_relu_dropout_bwd_colsum_f32(gy.data_ptr(), y.data_ptr(), gh.data_ptr(), out.data_ptr(), partials.data_ptr(), rows,
                                                    float(p), raw_stream())
    if code:
        raise RuntimeError("mono_relu_dropout_bwd_colsum_f32 failed with code %d" % code)
    return gh, out


def relu_dropout_backward(gy, y, p):
    gy = same_layout(y, gy)
    gh = torch.empty_like(y)
    with on_device(y.device):
        code = load().mono_relu_dropout_bwd_f32(gy.data_ptr(), y.data_ptr(), gh.data_ptr(), y.numel(), float(p), raw_stream())
    if code:
        raise RuntimeError("mono_relu_dropout_bwd_f32 failed with code %d" % code)
    return gh


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, p):
        y = relu_dropout_forward(h, p)
        ctx.save_for_backward(y)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return relu_dropout_backward(gy, y, ctx.p), None


RELU_DROPOUT_MIN_NUMEL = 1 << 20     # (the 8,800-token decoder FFNs included: 2 launches fewer each way per layer)


def relu_dropout(h, dropout):
    """``dropout(relu(h))`` for an ``nn.Dropout``: one HIP pass forward and one backward on large float32 GPU tensors in
    training (PyTorch: ReLU, dropout + mask, masked-scale, threshold-backward passes); PyTorch ops otherwise."""
    if h.is_cuda and h.dtype == torch.float32 and dropout.training and 0.0 < dropout.p < 1.0 and h.numel() % 4 == 0 \
            and h.numel() >= RELU_DROPOUT_MIN_NUMEL and torch.is_grad_enabled():
        return _ReluDropout.apply(h, dropout.p)
    return dropout(torch.relu(h))


# ---------------------------------------------------------------------------------------------------------
class _MatchedLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, boxes, depth, dims, angle, idx, t_box, t_depth, t_size, t_bin, t_res):
        NL, B, Q, _ = boxes.shape
        K = idx.size(2)
        out = torch.empty((NL, 6), dtype=torch.float32, device=boxes.device)
        comp = torch.empty(NL, dtype=torch.float32, device=boxes.device)
        tensors = (boxes, depth, dims, angle, idx, t_box, t_depth, t_size, t_bin, t_res)
        with on_device(boxes.device):
            code = load().mono_matched_losses_fwd_f32(*[t.data_ptr() for t in tensors], out.data_ptr(), comp.data_ptr(), NL, B, Q, K,
                                                      raw_stream())
        if code:
            raise RuntimeError("mono_matched_losses_fwd_f32 failed with code %d" % code)
        ctx.save_for_backward(*tensors, comp)
        return out

    @staticmethod
    def backward(ctx, go):
        *tensors, comp = ctx.saved_tensors
        boxes, depth, dims, angle = tensors[:4]
        NL, B, Q, _ = boxes.shape
        K = tensors[4].size(2)
        buf = torch.zeros((NL, B, Q, 35), dtype=torch.float32, device=boxes.device)       # one memset for the four gradients
        flat = buf.view(-1)
        n = NL * B * Q
        g_boxes, g_depth = flat[:n * 6].view(NL, B, Q, 6), flat[n * 6:n * 8].view(NL, B, Q, 2)
        g_dims, g_angle = flat[n * 8:n * 11].view(NL, B, Q, 3), flat[n * 11:].view(NL, B, Q, 24)
        go = go.contiguous()
        with on_device(boxes.device):
            code = load().mono_matched_losses_bwd_f32(*[t.data_ptr() for t in tensors], comp.data_ptr(), go.data_ptr(),
                                                      g_boxes.data_ptr(), g_depth.data_ptr(), g_dims.data_ptr(), g_angle.data_ptr(),
                                                      NL, B, Q, K, raw_stream())
        if code:
            raise RuntimeError("mono_matched_losses_bwd_f32 failed with code %d" % code)
        return (g_boxes, g_depth, g_dims, g_angle) + (None,) * 6


def matched_losses_supported(boxes, idx):
    return boxes.is_cuda and boxes.dtype == torch.float32 and idx.size(2) > 0


def matched_losses(boxes, depth, dims, angle, idx, t_box, t_depth, t_size, t_bin, t_res):
    """Per-layer sums [NL, 6] of {center, bbox, giou, depth, dim, angle} over the matched pairs ``idx`` [3, NL, K]."""
    c = lambda t, dt: t.to(dt).contiguous()
    f = torch.float32
    return _MatchedLosses.apply(c(boxes, f), c(depth, f), c(dims, f), c(angle, f), c(idx, torch.int64), c(t_box, f),
                                c(t_depth.reshape(-1), f), c(t_size, f), c(t_bin.reshape(-1), torch.int64), c(t_res.reshape(-1), f))


# ---- DDN depth-map loss (csrc/ddn_loss.hip) ---------------------------------------------------------------------------
def _ddn_strides(logits):
    """(batch, channel, pixel) strides in floats of a [B, C, H, W] map whose pixels are row-major with one stride:
    NCHW-contiguous and channels-last both qualify."""
    sb, sc, sh, sw = logits.stride()
    return (sb, sc, sw) if sh == sw * logits.shape[3] else None


def ddn_loss_supported(logits, boxes, depth, valid):
    return (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and _ddn_strides(logits) is not None
            and boxes.dtype == torch.float32 and depth.dtype == torch.float32 and valid.dtype == torch.bool)


DDN_EAGER_BACKWARD = True     # see _DDNLoss.forward


class _DDNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, boxes, depth, valid, alpha, gamma, fg_weight, bg_weight, depth_min, depth_max):
        B, C, H, W = logits.shape
        N = boxes.shape[1]
        boxes, depth, valid = boxes.contiguous(), depth.contiguous(), valid.contiguous()
        sb, sc, sp = _ddn_strides(logits)
        lib = load()
        partial = torch.empty(lib.mono_ddn_loss_blocks(B, H, W), dtype=torch.float32, device=logits.device)
        code = lib.mono_ddn_loss_fwd_f32(logits.data_ptr(), boxes.data_ptr(), depth.data_ptr(), valid.data_ptr(), partial.data_ptr(),
                                         B, C, H, W, N, sb, sc, sp, alpha, gamma, fg_weight, bg_weight, depth_min, depth_max,
                                         raw_stream())
        if code:
            raise RuntimeError("mono_ddn_loss_fwd_f32 failed with code %d" % code)
        ctx.consts = (alpha, gamma, fg_weight, bg_weight, depth_min, depth_max)
        ctx.eager = None
        if DDN_EAGER_BACKWARD and logits.requires_grad:
            # the gradient for an upstream factor of 1, evaluated NOW: the criterion calls this between the two halves of the
            # matcher, where the GPU would otherwise idle while the host solves the assignments -- 0.16 ms of the backward
            # moved under that wait; backward() only scales it
            one = torch.ones(1, dtype=torch.float32, device=logits.device)
            grad = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
            code = lib.mono_ddn_loss_bwd_f32(logits.data_ptr(), boxes.data_ptr(), depth.data_ptr(), valid.data_ptr(), one.data_ptr(),
                                             grad.data_ptr(), B, C, H, W, N, sb, sc, sp, *ctx.consts, raw_stream())
            if code:
                raise RuntimeError("mono_ddn_loss_bwd_f32 failed with code %d" % code)
            ctx.save_for_backward(grad)
            ctx.eager = True
        else:
            ctx.save_for_backward(logits, boxes, depth, valid)
        return partial.sum() / (B * H * W)

    @staticmethod
    def backward(ctx, g):
        if ctx.eager:
            (grad,) = ctx.saved_tensors
            return (grad * g,) + (None,) * 9
        logits, boxes, depth, valid = ctx.saved_tensors
        B, C, H, W = logits.shape
        sb, sc, sp = _ddn_strides(logits)
        grad = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
        g = g.reshape(1).to(torch.float32).contiguous()
        code = load().mono_ddn_loss_bwd_f32(logits.data_ptr(), boxes.data_ptr(), depth.data_ptr(), valid.data_ptr(), g.data_ptr(),
                                            grad.data_ptr(), B, C, H, W, boxes.shape[1], sb, sc, sp, *ctx.consts,
                                            raw_stream())
        if code:
            raise RuntimeError("mono_ddn_loss_bwd_f32 failed with code %d" % code)
        return (grad,) + (None,) * 9


def ddn_loss(logits, boxes, depth, valid, alpha, gamma, fg_weight, bg_weight, depth_min=1e-3, depth_max=60.0):
    return _DDNLoss.apply(logits, boxes, depth, valid, float(alpha), float(gamma), float(fg_weight), float(bg_weight),
                          float(depth_min), float(depth_max))


# ---- expected depth over the bin distribution (csrc/ddn_loss.hip) ------------------------------------------------------------
class _DepthExpectation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values):
        B, C, H, W = logits.shape
        sb, sc, sp = _ddn_strides(logits)
        values = values.contiguous()
        out = torch.empty((B, H, W), dtype=torch.float32, device=logits.device)
        code = load().mono_depth_expect_fwd_f32(logits.data_ptr(), values.data_ptr(), out.data_ptr(), B, C, H, W, sb, sc, sp,
                                                raw_stream())
        if code:
            raise RuntimeError("mono_depth_expect_fwd_f32 failed with code %d" % code)
        ctx.save_for_backward(logits, values, out)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, values, out = ctx.saved_tensors
        B, C, H, W = logits.shape
        sb, sc, sp = _ddn_strides(logits)
        grad = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
        code = load().mono_depth_expect_bwd_f32(logits.data_ptr(), values.data_ptr(), out.data_ptr(), g.contiguous().data_ptr(),
                                                grad.data_ptr(), B, C, H, W, sb, sc, sp, raw_stream())
        if code:
            raise RuntimeError("mono_depth_expect_bwd_f32 failed with code %d" % code)
        return grad, None


def depth_expectation(logits, values):
    """``(softmax(logits, 1) * values.view(1, -1, 1, 1)).sum(1)`` (depth_predictor.py:90-91) -- one HIP kernel each way for
    float32 CUDA logits [B, C, H, W] (NCHW or channels-last), the PyTorch expression otherwise."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and _ddn_strides(logits) is not None \
            and values.dtype == torch.float32 and not values.requires_grad:
        return _DepthExpectation.apply(logits, values)
    return (torch.softmax(logits, dim=1) * values.reshape(1, -1, 1, 1)).sum(dim=1)


# ---- classification side of the criterion (csrc/matched_losses.hip) ------------------------------------------------------------
class _FocalClassification(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, idx, labels, sizes, alpha, gamma):
        NL, B, Q, C = logits.shape
        K = idx.size(2)
        out = torch.empty((NL, 3), dtype=torch.float32, device=logits.device)
        with on_device(logits.device):
            code = load().mono_focal_fwd_f32(logits.data_ptr(), idx.data_ptr(), labels.data_ptr(), sizes.data_ptr(), out.data_ptr(),
                                             NL, B, Q, C, K, alpha, gamma, raw_stream())
        if code:
            raise RuntimeError("mono_focal_fwd_f32 failed with code %d" % code)
        ctx.save_for_backward(logits, idx, labels)
        ctx.consts = (alpha, gamma)
        return out

    @staticmethod
    def backward(ctx, go):
        logits, idx, labels = ctx.saved_tensors
        NL, B, Q, C = logits.shape
        g = go[:, 0].contiguous()                       # class / cardinality errors carry no gradient
        grad = torch.empty_like(logits)
        with on_device(logits.device):
            code = load().mono_focal_bwd_f32(logits.data_ptr(), idx.data_ptr(), labels.data_ptr(), g.data_ptr(), grad.data_ptr(),
                                             NL, B, Q, C, idx.size(2), *ctx.consts, raw_stream())
        if code:
            raise RuntimeError("mono_focal_bwd_f32 failed with code %d" % code)
        return grad, None, None, None, None, None


def focal_classification_supported(logits, idx):
    NL, B, Q, C = logits.shape
    return logits.is_cuda and logits.dtype == torch.float32 and C <= 255 and B <= 256 and B * Q <= 32768 and idx.size(2) > 0


def focal_classification(logits, idx, labels, sizes, alpha, gamma=2.0):
    """-> [NL, 3]: per decoder layer the sigmoid-focal-loss SUM against the matched one-hot targets (differentiable), the
    class error in % and the cardinality error (monodetr.py:396-449) -- one HIP launch each way."""
    return _FocalClassification.apply(logits.contiguous(), idx.contiguous(), labels.to(torch.int64).contiguous(),
                                      sizes.to(torch.float32).contiguous(), float(alpha), float(gamma))


# ---- the matcher's cost blocks (csrc/matched_losses.hip) ---------------------------------------------------------------------------
def match_cost_supported(logits, boxes):
    return logits.is_cuda and logits.dtype == torch.float32 and boxes.dtype == torch.float32 and boxes.shape[-1] == 6 and logits.dim() == 4


@torch.no_grad()
def match_cost_blocks(logits, boxes, labels, tboxes, cols, w_class, w_3d, w_bbox, w_giou):
    """[NL, B, Q, N] cost of every query against its image's own targets (matcher.py:53-88 per image block): one launch, the
    floats of the PyTorch expressions."""
    NL, B, Q, C = logits.shape
    N = cols.size(1)
    logits, boxes = logits.contiguous(), boxes.contiguous()
    labels, tboxes = labels.to(torch.int64).contiguous(), tboxes.to(torch.float32).contiguous()
    cols = cols.to(torch.int64).contiguous()
    out = torch.empty((NL, B, Q, N), dtype=torch.float32, device=logits.device)
    with on_device(logits.device):
        code = load().mono_match_cost_f32(logits.data_ptr(), boxes.data_ptr(), labels.data_ptr(), tboxes.data_ptr(), cols.data_ptr(),
                                          out.data_ptr(), NL, B, Q, C, N, float(w_class), float(w_3d), float(w_bbox), float(w_giou),
                                          raw_stream())
    if code:
        raise RuntimeError("mono_match_cost_f32 failed with code %d" % code)
    return out


# ---- the matcher's assignments on the device (csrc/lsap_device.hip) ------------------------------------------------------------------
LSAP_MAX_DIM, LSAP_MAX_CELLS = 128, 8192


def device_lsap_supported(blocks, sizes, group_num):
    """The device solver's limits (include/monosowa_pointwise.h); anything else stays on the host solver."""
    if not (blocks.is_cuda and blocks.dtype == torch.float32 and blocks.dim() == 4 and blocks.is_contiguous()):
        return False
    Q = blocks.shape[2]
    if group_num <= 0 or Q % group_num:
        return False
    gq, n = Q // group_num, max(sizes) if len(sizes) else 0
    return gq <= LSAP_MAX_DIM and n <= LSAP_MAX_DIM and min(gq, n) * max(gq, n) <= LSAP_MAX_CELLS and n <= blocks.shape[3]


@torch.no_grad()
def device_lsap_match_flat(blocks, sizes, group_num, status):
    """blocks [NL, B, Q, N] float32 (``match_cost_blocks``) -> the criterion's flat index tensor [3, NL, K] int64 ON THE DEVICE
    (image, query, flat target): the assignments of ``lsap.match_flat`` / scipy without a host round trip.  ``status``: a
    device int32 scalar the kernel ORs its flags into (1: invalid / infeasible cost matrix, 2: unsupported shape)."""
    NL, B, Q, N = blocks.shape
    gq = Q // group_num
    per = [group_num * min(gq, int(n)) for n in sizes]
    first = np.concatenate([[0], np.cumsum(per)[:-1]])
    toff = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    K = int(sum(per))
    out = torch.empty((3, NL, K), dtype=torch.int64, device=blocks.device)
    if K == 0:
        return out
    meta = torch.from_numpy(np.stack([np.asarray(sizes), first, toff]).astype(np.int32)).to(blocks.device, non_blocking=True)
    with on_device(blocks.device):
        code = load().mono_lsap_match_flat_f32(blocks.data_ptr(), NL, B, Q, N, group_num, meta.data_ptr(), out.data_ptr(), K,
                                               status.data_ptr(), raw_stream())
    if code:
        raise RuntimeError("mono_lsap_match_flat_f32 failed with code %d" % code)
    return out


# ---- per-level tail of the detection heads (csrc/head_tail.hip) -------------------------------------------------------------------
class _HeadTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tmp, size3d, depth_reg, wdepth, fu, img_h, ref):
        B, Q, _ = tmp.shape
        H, W = wdepth.shape[-2:]
        coords = torch.empty((B, Q, 6), dtype=torch.float32, device=tmp.device)
        dave = torch.empty((B, Q, 2), dtype=torch.float32, device=tmp.device)
        rp, rd = (ref.data_ptr(), ref.shape[-1]) if ref is not None else (None, 0)
        with on_device(tmp.device):
            code = load().mono_head_tail_fwd_f32(tmp.data_ptr(), size3d.data_ptr(), depth_reg.data_ptr(), wdepth.data_ptr(), fu.data_ptr(),
                                                 img_h.data_ptr(), coords.data_ptr(), dave.data_ptr(), B, Q, H, W, rp, rd, raw_stream())
        if code:
            raise RuntimeError("mono_head_tail_fwd_f32 failed with code %d" % code)
        ctx.save_for_backward(tmp, size3d, depth_reg, wdepth, fu, img_h, ref)
        return coords, dave

    @staticmethod
    def backward(ctx, g_coords, g_dave):
        tmp, size3d, depth_reg, wdepth, fu, img_h, ref = ctx.saved_tensors
        B, Q, _ = tmp.shape
        H, W = wdepth.shape[-2:]
        g_tmp, g_size, g_dreg = torch.empty_like(tmp), torch.empty_like(size3d), torch.empty_like(depth_reg)
        g_wd = torch.zeros_like(wdepth)
        gc = g_coords.contiguous() if g_coords is not None else None
        gd = g_dave.contiguous() if g_dave is not None else None
        rp, rd = (ref.data_ptr(), ref.shape[-1]) if ref is not None else (None, 0)
        with on_device(tmp.device):
            code = load().mono_head_tail_bwd_f32(tmp.data_ptr(), size3d.data_ptr(), depth_reg.data_ptr(), wdepth.data_ptr(), fu.data_ptr(),
                                                 img_h.data_ptr(), gc.data_ptr() if gc is not None else None,
                                                 gd.data_ptr() if gd is not None else None, g_tmp.data_ptr(), g_size.data_ptr(),
                                                 g_dreg.data_ptr(), g_wd.data_ptr(), B, Q, H, W, rp, rd, raw_stream())
        if code:
            raise RuntimeError("mono_head_tail_bwd_f32 failed with code %d" % code)
        return g_tmp, g_size, g_dreg, g_wd, None, None, None


def head_tail_supported(tmp, size3d, depth_reg, wdepth, fu, img_h):
    f = torch.float32
    return tmp.is_cuda and all(t.dtype == f for t in (tmp, size3d, depth_reg, wdepth, fu, img_h)) and tmp.shape[-1] == 6 \
        and size3d.shape[-1] == 3 and depth_reg.shape[-1] == 2 and wdepth.dim() == 3 and not fu.requires_grad and not img_h.requires_grad


def head_tail(tmp, size3d, depth_reg, wdepth, fu, img_h, ref=None):
    """-> coords [B, Q, 6] = sigmoid(tmp), depth_ave [B, Q, 2] (regressed + geometric + depth-map depth averaged, log-variance):
    monodetr.py:238-263 in one launch each way.  ``ref`` [B, Q, 2 | 6] (detached reference boxes): the logits are
    ``tmp + inverse_sigmoid(ref)`` on ref's coordinates (monodetr.py:224-232) -- the 7 launches of that expression folded in."""
    c = lambda t: t.contiguous()
    if ref is not None:
        assert not ref.requires_grad and ref.dtype == torch.float32 and ref.shape[:2] == tmp.shape[:2]
        ref = c(ref)
    return _HeadTail.apply(c(tmp), c(size3d), c(depth_reg), c(wdepth), c(fu.reshape(-1)), c(img_h.reshape(-1)), ref)


@torch.no_grad()
def refine_reference(tmp, ref):
    """sigmoid(tmp + inverse_sigmoid(ref) on ref's coordinates) -> [..., 6], no autograd: the decoder's reference refinement, which
    the reference detaches (depthaware_transformer.py:602-613); ~9 launches as one."""
    tmp, ref = tmp.detach().contiguous(), ref.detach().contiguous()
    out = torch.empty_like(tmp)
    n = tmp.numel() // 6
    with on_device(tmp.device):
        code = load().mono_refine_reference_f32(tmp.data_ptr(), ref.data_ptr(), out.data_ptr(), n, ref.shape[-1], raw_stream())
    if code:
        raise RuntimeError("mono_refine_reference_f32 failed with code %d" % code)
    return out


def refine_reference_supported(tmp, ref):
    return tmp.is_cuda and tmp.dtype == torch.float32 and ref.dtype == torch.float32 and tmp.shape[-1] == 6 and ref.shape[-1] in (2, 6) \
        and tmp.shape[:-1] == ref.shape[:-1]
