"""ResNet-50/101 body with frozen batch-norm -> C3/C4/C5 feature maps + sine position encodings.

Reference: lib/models/monodetr/backbone.py (FrozenBatchNorm2d :28-65, BackboneBase :68-91,
Backbone :94-115, Joiner :118-135, build_backbone :138-144).  The reference takes the network
from ``torchvision.models`` (pinned 0.14.1, not vendored, not installed here), so the ResNet-v1.5
bottleneck network is restated locally with torchvision's parameter names: a reference checkpoint
key such as ``backbone.0.body.layer3.5.conv2.weight`` loads unchanged.  Backbone parity against
torchvision is UNPINNED in this container.

MI355X notes: frozen BN is a per-channel affine, so each conv is run with the affine folded into
its weights/bias (one pass over the activation instead of two; exact in real arithmetic, ~1e-7 in
fp32) and the fold is differentiated through for the trainable stages.  The dense convolutions go
to MIOpen (MFMA); nothing here is hand-written.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..pointwise import (affine_relu, affine_relu_supported, bias_act, bias_act_fork, bias_relu_maxpool,
                         bias_relu_maxpool_supported, conv1x1_head, conv1x1_head_supported, conv1x1_tail,
                         conv1x1_tail_ds, conv1x1_tail_ds_supported, conv1x1_tail_supported)
from .misc import NestedTensor
from .position_encoding import build_position_encoding


AFFINE_IN_KERNEL = True      # frozen-BN affine map inside the ReLU kernels for residual-free trainable convolutions
CACHE_SCALE_SHIFT = True


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (buffers, as in the reference)."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def scale_shift(self):
        """Per-channel (scale, shift) of the frozen affine map.  The four buffers never change during training, so the
        pair is computed once and reused until a buffer is written (load_state_dict, .to(), in-place edits bump the
        tensors' version / identity): 53 layers x 4 tiny kernels per step otherwise."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b.data_ptr(), b._version, b.device, b.dtype) for b in bufs)
        cached = self.__dict__.get("_scale_shift")
        if CACHE_SCALE_SHIFT and cached is not None and cached[0] == key:
            return cached[1], cached[2]
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        self.__dict__["_scale_shift"] = (key, scale, shift)
        return scale, shift

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


# Optimizers may update parameters through raw pointers (the fused AdamW kernel does) without touching the tensors'
# version counters: every optimizer step anywhere invalidates the folded-weight cache below.
_PARAM_EPOCH = [0]


def _bump_param_epoch(*_args, **_kwargs):
    _PARAM_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(_bump_param_epoch)
except ImportError:             # very old torch: no global hook -> never cache
    _PARAM_EPOCH = None


def folded_weight(conv, bn, scale):
    """``conv.weight * scale`` (frozen BN folded into the convolution).  When no gradient will flow to the weight (frozen
    stem / layer1, or inference) the product only changes when the weight or the BN buffers do: kept across calls."""
    w = conv.weight
    if (w.requires_grad and torch.is_grad_enabled()) or _PARAM_EPOCH is None:
        return w * scale.view(-1, 1, 1, 1)
    # (a FROZEN weight is in no optimizer's parameter groups -- the raw-pointer AdamW never touches it -- so its folded product
    # outlives optimizer steps: with the epoch in its key the frozen stem / layer1 were re-folded and re-transposed every train step)
    key = (w.data_ptr(), w._version, scale.data_ptr(), scale._version, _PARAM_EPOCH[0] if w.requires_grad else -1)
    cached = conv.__dict__.get("_folded")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            cached = conv.__dict__["_folded"] = (key, (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
                                                 if w.is_contiguous(memory_format=torch.channels_last) else w * scale.view(-1, 1, 1, 1))
    return cached[1]


def _summed_shift(bn, other):
    """shift(bn) + shift(other) of two frozen norms (constant during training: kept)."""
    a, b = bn.scale_shift()[1], other.scale_shift()[1]
    key = (a.data_ptr(), a._version, b.data_ptr(), b._version)
    cached = bn.__dict__.get("_summed_shift")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            cached = bn.__dict__["_summed_shift"] = (key, a + b)
    return cached[1]


# Trainable 1 x 1 convolutions (stride 1: conv1 / conv3 of every bottleneck of layer2-4) with the frozen norm's scale and shift, the
# identity and the ReLU inside the GEMM's EPILOGUE (monosowa_amd/gemm_lt.py: hipBLASLt through the C-ABI shim, include/monosowa_gemm.h):
#     y[N H W, out] = relu(scale * (x[N H W, in] w^T) + identity + shift)            one launch over the channels-last pixel matrix
# instead of MIOpen's convolution followed by a pass over its output (bias_relu_mask / affine_relu_mask: 1.4 ms per step at B = 16).
# The backward reads the ReLU mask from y itself (relu_grad / relu_grad2 kernels), forms dX = (g * scale) w by a library GEMM and
# leaves the weight gradient with MIOpen's solver (round 5 A/B: a split-K GEMM for it was 3.4 ms per step SLOWER).
# Bits (A/B: tools/ab_step.py monosowa_amd.monodetr.backbone.CONV1X1_EPILOGUE 0 3): 1 forward through the epilogue GEMM, 2 input
# gradient through the library GEMM (else MIOpen's backward-data solver).
CONV1X1_EPILOGUE = int(os.environ.get("MONOSOWA_CONV1X1_EPILOGUE", "3"))
CONV1X1_SCALED_GRAD = 1      # nodes without an identity branch: the frozen norm's scale inside the ReLU backward's pass (2 launches fewer)


def _pixels(t):
    """[N, C, H, W] channels-last -> its [N H W, C] row-major matrix (a view)."""
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


class _Conv1x1BnAct(torch.autograd.Function):
    """relu(scale * conv1x1(x, w) + shift (+ residual)), returned n_out times (one tensor object per consumer, as bias_act_fork)."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, n_out):
        from .. import gemm_lt
        N, C, H, W = x.shape
        K = w.shape[0]
        y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        gemm_lt.gemm_nt(_pixels(x), w.reshape(K, C), scale, shift, None if residual is None else _pixels(residual), True, out=_pixels(y))
        ctx.save_for_backward(x, w, scale, y)
        ctx.has_res = residual is not None
        return (y,) + tuple(y.detach() for _ in range(n_out - 1))

    @staticmethod
    def backward(ctx, *grads):
        from .. import gemm_lt
        from ..pointwise import relu_grad_from_output
        x, w, scale, y = ctx.saved_tensors
        N, C, H, W = x.shape
        K = w.shape[0]
        given = [t for t in grads if t is not None]
        # Without an identity branch and with one consumer (conv1 of a bottleneck) the norm's scale goes onto the gradient inside the
        # ReLU backward's pass: dX = g' w and dW = g'^T x need neither a scaled copy of the weight nor a rescaled weight gradient.
        pre_scaled = bool(CONV1X1_SCALED_GRAD) and not ctx.has_res and len(given) == 1 and scale.data_ptr() % 16 == 0 and scale.is_contiguous()
        g = relu_grad_from_output(given, y, scale if pre_scaled else None)         # (sum of the consumers' gradients) * (y > 0), one pass
        gx = gw = None
        w2 = w.reshape(K, C)
        if ctx.needs_input_grad[0]:
            ws = w2 if pre_scaled else w2 * scale.view(-1, 1)                       # [out, in]: the scale rides on the (small) weight
            if CONV1X1_EPILOGUE & 2:
                gx = torch.empty_like(x)
                gemm_lt.gemm_nn(_pixels(g), ws, out=_pixels(gx))
            else:
                gx = torch.ops.aten.convolution_backward(g, x, ws.view(K, C, 1, 1), None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            # (MIOpen's weight-gradient solver: the library's own split-K TN GEMM for it -- gemm_lt.gemm_tn_bgrad, timed candidates --
            # measured +3.2 ms per step, PyTorch's sliced bmm +3.4 ms: round 5 A/Bs)
            gw = torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
            if not pre_scaled:
                gw = gw * scale.view(-1, 1, 1, 1)
        return gx, gw, None, None, (g if ctx.has_res else None), None


def conv1x1_no_grad(x, conv, scale, shift, residual, relu):
    """The same single launch where nothing asks for a gradient (inference, frozen stages without a fused kernel of their own):
    act(scale * conv1x1(x) + shift (+ residual)) -- or None when the path does not apply."""
    if not (CONV1X1_EPILOGUE & 1) or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1 or conv.padding != (0, 0) \
            or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last) \
            or x.shape[1] % 4 or conv.weight.shape[0] % 4 \
            or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or (residual is not None and residual.requires_grad))) \
            or (residual is not None and not (residual.shape[1] == conv.weight.shape[0] and residual.dtype == torch.float32
                                              and residual.is_contiguous(memory_format=torch.channels_last))):
        return None
    from .. import gemm_lt
    N, C, H, W = x.shape
    K = conv.weight.shape[0]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.no_grad():
        gemm_lt.gemm_nt(_pixels(x), conv.weight.reshape(K, C), scale, shift, None if residual is None else _pixels(residual), relu, out=_pixels(y))
    return y


def conv1x1_epilogue_applies(x, conv, residual=None):
    from .. import gemm_lt
    return bool(CONV1X1_EPILOGUE & 1) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.padding == (0, 0) \
        and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and torch.is_grad_enabled() and conv.weight.requires_grad \
        and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 4 == 0 and conv.weight.shape[0] % 4 == 0 \
        and (residual is None or (residual.shape[1] == conv.weight.shape[0] and residual.is_contiguous(memory_format=torch.channels_last)
                                  and residual.dtype == torch.float32))


def conv_bn_fork(x, conv, bn, residual, n_out=2, residual_bn=None):
    """``conv_bn(x, conv, bn, residual)`` returned as a pair for its two consumers (next block's first convolution and
    identity branch): their gradients are added inside the fused ReLU backward (``pointwise.bias_act_fork``).
    residual_bn: the residual is a downsample convolution's RAW output (scale folded into its weights, shift not yet added) -- its
    frozen norm's shift joins this norm's shift, so the downsample branch needs no pass of its own over its output."""
    if isinstance(bn, FrozenBatchNorm2d):
        scale, shift = bn.scale_shift()
        if residual_bn is not None:
            shift = _summed_shift(bn, residual_bn)
        if conv1x1_epilogue_applies(x, conv, residual):
            return _Conv1x1BnAct.apply(x, conv.weight, scale, shift, residual, n_out)
        y = conv1x1_no_grad(x, conv, scale, shift, residual, True)
        if y is not None:
            return (y,) * n_out
        y = F.conv2d(x, folded_weight(conv, bn, scale), None, conv.stride, conv.padding, conv.dilation, conv.groups)
        return bias_act_fork(y, shift, residual, n_out)
    assert residual_bn is None
    out = conv_bn(x, conv, bn, residual)
    return (out,) * n_out


def conv_bn(x, conv, bn, residual=None, relu=True):
    """relu(frozen_bn(conv(x)) (+ residual)): one convolution with the BN scale folded into the weights, then ONE
    in-place pointwise pass for shift, residual and ReLU (``monosowa_amd.pointwise``; ATen would issue a bias pass,
    an add pass and a ReLU pass)."""
    if isinstance(bn, FrozenBatchNorm2d):
        scale, shift = bn.scale_shift()
        if relu and residual is None and conv1x1_epilogue_applies(x, conv):
            return _Conv1x1BnAct.apply(x, conv.weight, scale, shift, None, 1)[0]
        if AFFINE_IN_KERNEL and relu and residual is None and conv.weight.requires_grad and torch.is_grad_enabled() and x.is_cuda:
            # trainable convolution without residual: raw weights, the BN affine map runs inside the ReLU kernels
            # (forward y*scale+shift, backward grad*mask*scale) -- no weight multiply per step, none in the backward
            y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if affine_relu_supported(y, scale):
                return affine_relu(y, scale, shift)
            return torch.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        y = conv1x1_no_grad(x, conv, scale, shift, residual, relu)
        if y is not None:
            return y
        y = F.conv2d(x, folded_weight(conv, bn, scale), None, conv.stride, conv.padding, conv.dilation, conv.groups)
        return bias_act(y, shift, residual, relu)
    y = bn(conv(x))
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


FOLD_DOWNSAMPLE_SHIFT = os.environ.get("MONOSOWA_FOLD_DS_SHIFT", "1") != "0"   # A/B switch (tools)
FUSED_FROZEN_DS = os.environ.get("MONOSOWA_FUSED_FROZEN_DS", "1") != "0"       # A/B switch (tools)
FUSED_FROZEN_TAIL = os.environ.get("MONOSOWA_FUSED_FROZEN_TAIL", "1") != "0"   # A/B switch (tools)
FUSED_STEM = os.environ.get("MONOSOWA_FUSED_STEM", "1") != "0"      # A/B switch (tools): 0 = in-place bias + ReLU pass, then F.max_pool2d


def stem(x, conv, bn):
    """max_pool(relu(bn1(conv1(x))), 3, 2, 1) (torchvision's ResNet stem behind backbone.py:83 of the reference).  The stem is frozen
    (backbone.py:72-74): when nothing asks for a gradient the BN shift, the ReLU and the pooling are one pass over the convolution's
    output (pointwise.bias_relu_maxpool) instead of an in-place pass plus a pooling pass over the largest activation of the network."""
    if isinstance(bn, FrozenBatchNorm2d):
        scale, shift = bn.scale_shift()
        w = folded_weight(conv, bn, scale)
        if not (w.requires_grad and torch.is_grad_enabled()):
            y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if FUSED_STEM and bias_relu_maxpool_supported(y, shift):
                return bias_relu_maxpool(y, shift)
            return F.max_pool2d(bias_act(y, shift, None, True), kernel_size=3, stride=2, padding=1)
    return F.max_pool2d(conv_bn(x, conv, bn), kernel_size=3, stride=2, padding=1)


class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck: 1x1 -> 3x3 (carries the stride) -> 1x1 (x4), residual add, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.downsample = downsample

    def forward(self, x):
        """x: a tensor, or the (a, b) pair a previous block returned (same values; one per consumer).  Returns a pair."""
        xa, xb = x if isinstance(x, tuple) else (x, x)
        n_out = getattr(self, "n_out", 2)
        whole = self._frozen_block(xa, xb, n_out) if FUSED_FROZEN_TAIL else None
        if whole is not None:
            return whole
        out = conv_bn(xa, self.conv1, self.bn1)
        out = conv_bn(out, self.conv2, self.bn2)
        if self.downsample is None:
            return conv_bn_fork(out, self.conv3, self.bn3, xb, n_out)
        ds_conv, ds_bn = self.downsample[0], self.downsample[1]
        if FOLD_DOWNSAMPLE_SHIFT and isinstance(ds_bn, FrozenBatchNorm2d) and isinstance(self.bn3, FrozenBatchNorm2d):
            # relu(y3 + shift3 + (y_ds + shift_ds)) = relu(y3 + (shift3 + shift_ds) + y_ds): the downsample's output is only ever this
            # residual, so its shift rides on bn3's and the branch is the bare convolution (no bias pass over its output)
            identity = F.conv2d(xb, folded_weight(ds_conv, ds_bn, ds_bn.scale_shift()[0]), None, ds_conv.stride, ds_conv.padding,
                                ds_conv.dilation, ds_conv.groups)
            return conv_bn_fork(out, self.conv3, self.bn3, identity, n_out, residual_bn=ds_bn)
        identity = conv_bn(xb, ds_conv, ds_bn, relu=False)
        return conv_bn_fork(out, self.conv3, self.bn3, identity, n_out)


    def _frozen_block(self, xa, xb, n_out):
        """The whole block when it is frozen (layer1: backbone.py:72-74 of the reference) and 64 -> 256 channels wide: conv1 + bn1 + ReLU
        is one pass (pointwise.conv1x1_head), the 3 x 3 convolution runs bare, and its shift + ReLU, conv3, bn3's shift, the identity and
        the final ReLU are ONE pass (pointwise.conv1x1_tail) instead of three over the largest activations of the network.
        None: not applicable."""
        bn1, bn2, bn3, ds = self.bn1, self.bn2, self.bn3, self.downsample
        if self.conv3.weight.shape[:2] != (256, 64) or not xa.is_cuda or not isinstance(bn2, FrozenBatchNorm2d) \
                or not isinstance(bn3, FrozenBatchNorm2d) or (ds is not None and not isinstance(ds[1], FrozenBatchNorm2d)):
            return None
        if torch.is_grad_enabled() and (xa.requires_grad or xb.requires_grad or self.conv1.weight.requires_grad
                                        or self.conv2.weight.requires_grad or self.conv3.weight.requires_grad
                                        or (ds is not None and ds[0].weight.requires_grad)):
            return None
        h1 = None
        conv1 = self.conv1
        # (the library's epilogue GEMM in place of conv1x1_head: +0.06 ms per step in the round-5 A/B -- the own kernel stays)
        if isinstance(bn1, FrozenBatchNorm2d) and conv1.kernel_size == (1, 1) and conv1.stride == (1, 1) and conv1.weight.shape[0] == 64:
            scale1, shift1 = bn1.scale_shift()
            w1 = folded_weight(conv1, bn1, scale1)
            cached = self.__dict__.get("_head_w")
            if cached is None or cached[0] is not w1:
                with torch.no_grad():
                    cached = self.__dict__["_head_w"] = (w1, w1.reshape(64, -1).t().contiguous())           # [in, out]
            if conv1x1_head_supported(xa, cached[1]):
                h1 = conv1x1_head(xa, cached[1], shift1)
        if h1 is None:
            h1 = conv_bn(xa, conv1, bn1)
        (scale2, shift2), (scale3, shift3) = bn2.scale_shift(), bn3.scale_shift()
        w3 = folded_weight(self.conv3, bn3, scale3)                       # frozen: the folded weight is a kept tensor
        cached = self.__dict__.get("_tail_w")
        if cached is None or cached[0] is not w3:
            with torch.no_grad():
                cached = self.__dict__["_tail_w"] = (w3, w3.reshape(256, 64).t().contiguous())      # [in, out]
        w_kn = cached[1]
        conv2 = self.conv2
        raw2 = F.conv2d(h1, folded_weight(conv2, bn2, scale2), None, conv2.stride, conv2.padding, conv2.dilation, conv2.groups)
        if ds is not None and FUSED_FROZEN_DS and ds[0].kernel_size == (1, 1) and ds[0].stride == (1, 1) and ds[0].weight.shape[:2] == (256, 64):
            # the stage's first block: the downsample product of the block's input joins conv3's accumulator (no identity tensor)
            wd = folded_weight(ds[0], ds[1], ds[1].scale_shift()[0])
            cached_d = self.__dict__.get("_tail_wd")
            if cached_d is None or cached_d[0] is not wd:
                with torch.no_grad():
                    cached_d = self.__dict__["_tail_wd"] = (wd, wd.reshape(256, 64).t().contiguous())
            if conv1x1_tail_ds_supported(raw2, w_kn, xb, cached_d[1]):
                return (conv1x1_tail_ds(raw2, shift2, w_kn, xb, cached_d[1], _summed_shift(bn3, ds[1])),) * n_out
        if ds is None:
            identity, b_out = xb, shift3
        else:
            identity = F.conv2d(xb, folded_weight(ds[0], ds[1], ds[1].scale_shift()[0]), None, ds[0].stride, ds[0].padding,
                                ds[0].dilation, ds[0].groups)
            b_out = _summed_shift(bn3, ds[1])
        if not conv1x1_tail_supported(raw2, w_kn, identity):
            # (not channels-last on this path, ...): the three passes on the tensors already computed
            out = bias_act(raw2, shift2, None, True)
            y = F.conv2d(out, w3, None, self.conv3.stride, self.conv3.padding, self.conv3.dilation, self.conv3.groups)
            return (bias_act(y, b_out, identity, True),) * n_out
        return (conv1x1_tail(raw2, shift2, w_kn, b_out, identity),) * n_out


_DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class ResNetBody(nn.Module):
    """conv1 .. layer4 with torchvision's child names; returns the outputs of layer2/3/4."""

    def __init__(self, name, dilation=False, norm_layer=FrozenBatchNorm2d, return_interm_layers=True):
        super().__init__()
        if name not in _DEPTHS:
            raise ValueError("backbone %r is not supported (resnet50 / resnet101)" % name)
        self.inplanes, self._dil = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.layer1 = self._make_layer(64, _DEPTHS[name][0], 1, False, norm_layer)
        self.layer2 = self._make_layer(128, _DEPTHS[name][1], 2, False, norm_layer)
        self.layer3 = self._make_layer(256, _DEPTHS[name][2], 2, False, norm_layer)
        self.layer4 = self._make_layer(512, _DEPTHS[name][3], 2, dilation, norm_layer)
        self.return_interm_layers = return_interm_layers
        for m in self.modules():   # torchvision's default init (pretrained weights are a download)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride, dilate, norm_layer):
        prev_dil = self._dil
        if dilate:
            self._dil *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       norm_layer(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, prev_dil, norm_layer)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes, dilation=self._dil, norm_layer=norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = stem(x, self.conv1, self.bn1)
        # blocks hand (a, b) pairs to each other (one tensor object per consumer, see Bottleneck.forward); the stage
        # outputs that leave the body are the first members
        # a stage output that also leaves the body (layer2 / layer3 with intermediate layers) has a third consumer: its last block
        # returns three handles, the third goes out; layer4's two handles serve the two projections that read c5 (the second
        # one rides along as `fork_twin`): no gradient accumulation pass over any stage output
        x = self.layer1(x)
        if self.return_interm_layers:
            self.layer2[-1].n_out = self.layer3[-1].n_out = 3
        c3 = self.layer2(x)
        c4 = self.layer3(c3[:2])
        c5 = self.layer4(c4[:2])
        out5 = c5[0]
        if c5[1] is not c5[0]:
            out5.fork_twin = c5[1]
        return {"0": c3[-1], "1": c4[-1], "2": out5} if self.return_interm_layers else {"0": out5}


class Backbone(nn.Module):
    """ResNet body with frozen BN; conv1 + layer1 frozen, layer2-4 trainable when ``train_backbone``
    (backbone.py:72-74)."""

    def __init__(self, name, train_backbone, return_interm_layers, dilation, depth=False, pretrained=False):
        super().__init__()
        if depth:
            raise NotImplementedError("4-channel (RGB-D) input is off in every shipped config (model.depth: False)")
        if pretrained:
            raise RuntimeError("model.pretrained: True needs torchvision's ImageNet download, which is not "
                               "available offline; load a state dict instead and set pretrained: False")
        assert name not in ("resnet18", "resnet34"), "number of channels are hard coded"
        self.body = ResNetBody(name, dilation, FrozenBatchNorm2d, return_interm_layers)
        for pname, p in self.body.named_parameters():
            if not train_backbone or ("layer2" not in pname and "layer3" not in pname and "layer4" not in pname):
                p.requires_grad_(False)
        if return_interm_layers:
            self.strides, self.num_channels = [8, 16, 32], [512, 1024, 2048]
        else:
            self.strides, self.num_channels = [32], [2048]
        if dilation:
            self.strides[-1] = self.strides[-1] // 2

    def forward(self, images):
        out = {}
        for name, x in self.body(images).items():
            mask = torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.bool, device=x.device)
            out[name] = NestedTensor(x, mask, all_valid=True)
        return out


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, images):
        xs = self[0](images)
        out = [x for _, x in sorted(xs.items())]
        pos = [self[1](x).to(x.tensors.dtype) for x in out]
        return out, pos


def build_backbone(cfg):
    position_embedding = build_position_encoding(cfg)
    return_interm_layers = cfg["masks"] or cfg["num_feature_levels"] > 1
    backbone = Backbone(cfg["backbone"], cfg["train_backbone"], return_interm_layers, cfg["dilation"],
                        cfg["depth"], cfg["pretrained"])
    return Joiner(backbone, position_embedding)
