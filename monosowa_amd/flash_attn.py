"""fp32 head_dim-32 attention through the C-ABI library (include/monosowa_attn.h): the scaled-dot-product core of
MonoDETR's depth-encoder self-attention and decoder depth cross-attention on the exact-f32 matrix cores."""
import ctypes
import math
import os

import torch

from . import flops
from ._lib import raw_stream, on_device
from .token_linear import linear as fast_linear

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_attn.so")
SYMBOLS = ("mono_attn_forward_f32", "mono_attn_backward_f32", "mono_attn_forward_masked_f32", "mono_attn_backward_masked_f32",
           "mono_attn_keep_words", "mono_attn_forward_keep_f32", "mono_attn_backward_keep_f32")
_lib = None


class _Strides(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_longlong), ("head", ctypes.c_longlong), ("token", ctypes.c_longlong)]


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("MONOSOWA_ATTN_LIB", _PATH)             # another BUILD of the same library (A/B measurements)
        if not os.path.exists(path):
            raise RuntimeError("HIP extension %s is missing: run `python -m monosowa_amd.build`" % path)
        lib = ctypes.CDLL(path)
        P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_ulonglong
        lib.mono_attn_forward_f32.restype = I
        lib.mono_attn_forward_f32.argtypes = [P] * 5 + [I] * 5 + [_Strides] * 4 + [F, F, U, P]
        lib.mono_attn_backward_f32.restype = I
        lib.mono_attn_backward_f32.argtypes = [P] * 10 + [I] * 5 + [_Strides] * 7 + [F, F, U, P]
        lib.mono_attn_forward_masked_f32.restype = I
        lib.mono_attn_forward_masked_f32.argtypes = [P] * 6 + [I] * 5 + [_Strides] * 4 + [F, F, U, P]
        lib.mono_attn_backward_masked_f32.restype = I
        lib.mono_attn_backward_masked_f32.argtypes = [P] * 11 + [I] * 5 + [_Strides] * 7 + [F, F, U, P]
        lib.mono_attn_keep_words.restype = ctypes.c_longlong
        lib.mono_attn_keep_words.argtypes = [I] * 4
        lib.mono_attn_forward_keep_f32.restype = I
        lib.mono_attn_forward_keep_f32.argtypes = [P] * 7 + [I] * 5 + [_Strides] * 4 + [F, F, U, P]
        lib.mono_attn_backward_keep_f32.restype = I
        lib.mono_attn_backward_keep_f32.argtypes = [P] * 12 + [I] * 5 + [_Strides] * 7 + [F, F, U, P]
        _lib = lib
    return _lib


def _strides(t):
    """t: [B, H, L, 32] view with a contiguous last dimension."""
    assert t.stride(3) == 1
    return _Strides(t.stride(0), t.stride(1), t.stride(2))


def supported(q, k, v):
    ok = lambda t: t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.size(3) == 32 and t.stride(3) == 1 \
        and t.data_ptr() % 16 == 0 and all(s % 4 == 0 for s in t.stride()[:3])
    return ok(q) and ok(k) and ok(v) and q.size(0) * q.size(1) <= 65535 and k.shape == v.shape \
        and q.shape[:2] == k.shape[:2]


_seed_counter = [0]
# the training forward hands its dropout mask to the backward as 1 bit per score (59 MB for the 1920 x 1920 depth-encoder call at B = 16)
# instead of the backward re-hashing it; MONOSOWA_ATTN_KEEP_BITS=0: regenerate (the same mask)
SAVE_KEEP_BITS = os.environ.get("MONOSOWA_ATTN_KEEP_BITS", "1") != "0"


def _next_seed():
    _seed_counter[0] += 1
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0] * 0xD1B54A32D192ED03 + rank * 0x94D049BB133111EB) & (2 ** 63 - 1)   # fits int64: autograd / profiler argument records


def _mask_ptr(mask, B, Lk):
    """key padding mask -> pointer of a contiguous [B, Lk] byte tensor (0 for None); non-zero / True = padded key"""
    if mask is None:
        return 0, None
    if mask.dtype not in (torch.bool, torch.uint8) or tuple(mask.shape) != (B, Lk) or not mask.is_cuda:
        raise RuntimeError("flash_attn: key_padding_mask must be a [B, Lk] bool / uint8 tensor on the GPU (True = padding); "
                           "additive float masks take the nn.MultiheadAttention path (mha_supported)")
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    mask = mask.contiguous()
    return mask.data_ptr(), mask


def keep_bits_like(q, k, p):
    """The buffer the forward leaves its dropout keep bits in for the backward (None without dropout): include/monosowa_attn.h."""
    if not p > 0:
        return None
    B, H, Lq, _ = q.shape
    return torch.empty(load().mono_attn_keep_words(B, H, Lq, k.size(2)), dtype=torch.int32, device=q.device)


def forward(q, k, v, scale, p, seed, key_padding_mask=None, keep_bits=None):
    """-> (o, lse); keep_bits (``keep_bits_like``): also filled with the dropout mask, for ``backward(..., keep_bits=)``."""
    B, H, Lq, _ = q.shape
    o = torch.empty((B, H, Lq, 32), dtype=torch.float32, device=q.device)
    lse = torch.empty((B * H, Lq), dtype=torch.float32, device=q.device)
    mp, keep = _mask_ptr(key_padding_mask, B, k.size(2))
    flops.attention_forward(B, H, Lq, k.size(2))
    with on_device(q.device):
        code = load().mono_attn_forward_keep_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, keep_bits.data_ptr() if keep_bits is not None else 0,
                                                 o.data_ptr(), lse.data_ptr(), B, H, Lq, k.size(2), 32, _strides(q), _strides(k), _strides(v),
                                                 _strides(o), float(scale), float(p), seed, raw_stream())
    if code:
        raise RuntimeError("mono_attn_forward_keep_f32 failed with code %d" % code)
    return o, lse


def _batch_major(t):
    """[B, H, L, 32] view whose batch stride is the outermost: a [B, L, H*32] buffer (else [L, B, H*32])"""
    return t.stride(0) >= t.stride(2)


def _like_heads(t):
    """A fresh dense [B, H, L, 32] view in the memory order of ``t`` -- [L, B, H*32] or [B, L, H*32] -- so that the gradient of
    a projection output lies like the output itself and flows back into its GEMMs without a layout copy."""
    B, H, L, _ = t.shape
    if _batch_major(t):
        return torch.empty((B, L, H * 32), dtype=torch.float32, device=t.device).view(B, L, H, 32).permute(0, 2, 1, 3)
    return torch.empty((L, B, H * 32), dtype=torch.float32, device=t.device).view(L, B, H, 32).permute(1, 2, 0, 3)


def backward(q, k, v, o, lse, dout, scale, p, seed, key_padding_mask=None, keep_bits=None):
    """dq, dk, dv as [B, H, L, 32] views of fresh buffers laid out like q, k, v (the layout of the MHA projections).
    keep_bits: the forward's dropout mask (else it is regenerated from the seed: the same mask, 14 % more backward time)."""
    B, H, Lq, _ = q.shape
    Lk = k.size(2)
    dq, dk, dv = _like_heads(q), _like_heads(k), _like_heads(v)
    delta = torch.empty((B * H, Lq), dtype=torch.float32, device=q.device)
    assert dout.stride() == o.stride()
    mp, keep = _mask_ptr(key_padding_mask, B, Lk)
    flops.attention_backward(B, H, Lq, Lk)
    with on_device(q.device):
        code = load().mono_attn_backward_keep_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, keep_bits.data_ptr() if keep_bits is not None else 0,
                                                  o.data_ptr(), lse.data_ptr(), dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                  delta.data_ptr(), B, H, Lq, Lk, 32, _strides(q), _strides(k), _strides(v), _strides(o),
                                                  _strides(dq), _strides(dk), _strides(dv), float(scale), float(p), seed, raw_stream())
    if code:
        raise RuntimeError("mono_attn_backward_keep_f32 failed with code %d" % code)
    return dq, dk, dv


class _Attention(torch.autograd.Function):
    """q, k, v: [B, H, L, 32] (strided views are fine); returns O as a [B, H, Lq, 32] view of an [Lq, B, H*32] buffer."""

    @staticmethod
    def forward(ctx, q, k, v, scale, p, seed, mask=None):
        B, H, Lq, _ = q.shape
        o = _like_heads(q)                           # [Lq, B, H*32] or, for a batch-major q, [B, Lq, H*32]
        lse = torch.empty((B * H, Lq), dtype=torch.float32, device=q.device)
        mp, mask = _mask_ptr(mask, B, k.size(2))
        bits = keep_bits_like(q, k, p) if SAVE_KEEP_BITS else None      # 1 bit per score, kept until the backward
        flops.attention_forward(B, H, Lq, k.size(2))
        with on_device(q.device):
            code = load().mono_attn_forward_keep_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, bits.data_ptr() if bits is not None else 0,
                                                     o.data_ptr(), lse.data_ptr(), B, H, Lq, k.size(2), 32, _strides(q), _strides(k),
                                                     _strides(v), _strides(o), float(scale), float(p), seed, raw_stream())
        if code:
            raise RuntimeError("mono_attn_forward_keep_f32 failed with code %d" % code)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.p, ctx.seed, ctx.mask, ctx.bits = scale, p, seed, mask, bits
        return o

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o, lse = ctx.saved_tensors
        if dout.stride() != o.stride():
            d2 = _like_heads(o)
            d2.copy_(dout)
            dout = d2
        dq, dk, dv = backward(q, k, v, o, lse, dout, ctx.scale, ctx.p, ctx.seed, ctx.mask, ctx.bits)
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, dropout_p=0.0, scale=None, seed=None, key_padding_mask=None):
    """softmax(q k^T * scale) with dropout, times v, per (batch, head); q/k/v [B, H, L, 32] float32 on the GPU.
    key_padding_mask: [B, Lk] bool / uint8, True = the key is padding (nn.MultiheadAttention's convention)."""
    if not supported(q, k, v):
        raise RuntimeError("flash_attn.attention: unsupported tensors (need float32 GPU [B,H,L,32], 16-byte aligned strides)")
    scale = 1.0 / math.sqrt(q.size(-1)) if scale is None else scale
    if dropout_p > 0 and seed is None:
        seed = _next_seed()
    return _Attention.apply(q, k, v, scale, float(dropout_p), int(seed or 0), key_padding_mask)


def mha_forward(mha, query, key, value, key_padding_mask=None):
    """``mha(query, key, value, key_padding_mask=..., need_weights=False)[0]`` for an ``nn.MultiheadAttention`` with 32-channel
    heads, [L, B, E] inputs (no attn_mask; the key padding mask -- [B, Lk], True = padding -- goes into the kernels): packed input projections, the HIP attention core on strided views (no head-major
    copies), output projection.  (torch.nn.functional.multi_head_attention_forward, as called at
    depth_predictor/transformer.py:59 and depthaware_transformer.py:417.)"""
    E, H = mha.embed_dim, mha.num_heads
    w, bias = mha.in_proj_weight, mha.in_proj_bias
    Lq, B, _ = query.shape
    Lk = key.size(0)
    def lin2(x, w_, b_):
        """two column blocks of one projection; for a batch-major input the split happens on the batch-major output, so that its
        backward concatenates the two gradients in the layout the GEMM's backward wants (no [B, L, 2 E] copy)"""
        if not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            a, b2 = fast_linear(x.transpose(0, 1), w_, b_).split(E, -1)
            return a.transpose(0, 1), b2.transpose(0, 1)
        return fast_linear(x, w_, b_).split(E, -1)

    def lin(x, w_, b_):
        # the [L, B, E] view of a batch-major buffer (the decoder's queries, tokens cut out of an NHWC map): project the buffer as
        # it lies -- a non-contiguous input costs F.linear a copy and an unfused bias add -- and hand back the same kind of view
        if not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            return fast_linear(x.transpose(0, 1), w_, b_).transpose(0, 1)
        return fast_linear(x, w_, b_)
    # row blocks of the packed projection and column blocks of a merged output come from ONE split each: its backward is a
    # single concatenation, where every `w[a:b]` slice would zero-fill a full-size gradient, copy its block in and have
    # autograd add the pieces (5 launches per pair; 15 per call over weights, biases and activations)
    if query is key:                                    # depth encoder: q = k = src + pos, v = src
        (w_qk, w_v), (b_qk, b_v) = w.split([2 * E, E]), bias.split([2 * E, E])
        q, k = lin2(query, w_qk, b_qk)
        v = lin(value, w_v, b_v)
    elif key is value:                                  # decoder: k = v = depth-aware tokens
        (w_q, w_kv), (b_q, b_kv) = w.split([E, 2 * E]), bias.split([E, 2 * E])
        q = lin(query, w_q, b_q)
        k, v = lin2(key, w_kv, b_kv)
    else:
        (w_q, w_k, w_v), (b_q, b_k, b_v) = w.split(E), bias.split(E)
        q, k, v = lin(query, w_q, b_q), lin(key, w_k, b_k), lin(value, w_v, b_v)
    heads = lambda t, L: t.unflatten(-1, (H, 32)).permute(1, 2, 0, 3)           # [L,B,E] -> [B,H,L,32] view
    o = attention(heads(q, Lq), heads(k, Lk), heads(v, Lk), mha.dropout if mha.training else 0.0, key_padding_mask=key_padding_mask)
    if _batch_major(o):                                 # o lies like q: out_proj on [B, Lq, E], returned as its [Lq, B, E] view
        return fast_linear(o.permute(0, 2, 1, 3).reshape(B, Lq, E), mha.out_proj.weight, mha.out_proj.bias).transpose(0, 1)
    return fast_linear(o.permute(2, 0, 1, 3).reshape(Lq, B, E), mha.out_proj.weight, mha.out_proj.bias)


def mask_supported(mask, B, Lk, device):
    """What the kernels read as a key padding mask: a [B, Lk] bool / uint8 tensor on the queries' device (non-zero = padding).
    nn.MultiheadAttention also takes an ADDITIVE float mask (0 / -inf): that one, a mask of another shape or on another device
    stays with the module path instead of being misread as bytes."""
    return mask is None or (torch.is_tensor(mask) and mask.dtype in (torch.bool, torch.uint8) and tuple(mask.shape) == (B, Lk)
                            and mask.device == device)


def mha_supported(mha, query, key, value, key_padding_mask=None):
    return (query.is_cuda and query.dtype == torch.float32 and mha.embed_dim == mha.num_heads * 32 and not mha.batch_first
            and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None and not mha.add_zero_attn
            and query.dim() == 3 and key.shape == value.shape and query.size(1) * mha.num_heads <= 65535
            and mask_supported(key_padding_mask, query.size(1), key.size(0), query.device))
