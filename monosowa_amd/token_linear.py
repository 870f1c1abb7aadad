"""Linear layer over very many tokens with a split-K weight gradient.

In the visual encoder every projection runs over B*S = 163,200 tokens (B=16, 1280x384) with 256 (or 128)
output features.  The forward and input-gradient GEMMs are tall and parallelise well, but the weight
gradient  dW[256,256] = dY^T[256,T] @ X[T,256]  has a 256x256 output and K = T = 163,200: the library kernel
runs it on 32 workgroups of a 256-CU chip (0.48 ms).  Splitting K into slices turns it into a batched
GEMM with 64x more workgroups followed by a tiny reduction (0.18 ms, `tools/gemm_probe.py`).
Same math as ``F.linear`` (fp32 summation order differs in the weight gradient: ~6e-6 relative).
"""
import torch
import torch.nn.functional as F

from .pointwise import colsum, sum_slices, linear_wgrad, linear_wgrad_applies

USE_SUM_SLICES = True     # the slices of a split-K weight gradient added by one coalesced pass (pointwise.sum_slices)
MIN_TOKENS = 32768      # split-K weight gradient from here on
SMALL_WGRAD_KERNEL = int(__import__('os').environ.get('MONOSOWA_SMALL_WGRAD', '1'))   # dW + db below MIN_TOKENS by csrc/small_wgrad.hip
MIN_ROWS = int(__import__('os').environ.get('MONOSOWA_TL_MIN_ROWS', '32768'))   # our backward from here on


def _slices(tokens):
    for s in (64, 48, 32, 24, 16, 8):
        if tokens % s == 0 and tokens // s >= 512:
            return s
    return 0


def weight_grad(g2, x2):
    """g2^T @ x2 for [tokens, out] / [tokens, in] matrices: split-K batched GEMM when there are many tokens."""
    s = _slices(x2.shape[0]) if x2.shape[0] >= MIN_TOKENS else 0
    if s:
        parts = torch.bmm(g2.view(s, -1, g2.shape[1]).transpose(1, 2), x2.view(s, -1, x2.shape[1]))
        return sum_slices(parts) if USE_SUM_SLICES else parts.sum(0)
    return small_weight_grad(g2, x2)


# (Round 5, measured and NOT used: hipBLASLt's bias-gradient epilogue -- dW = dY^T X with db = the column sums of dY from the same launch,
# gemm_lt.gemm_tn_bgrad through the C-ABI shim -- is correct (tests/test_gemm_lt_gpu.py) and slow: ~1 ms per call at [8800, 256] x
# [8800, 256], +45 ms per train step when every token linear below MIN_TOKENS rows used it, tools/ab_step.py.  The column-sum kernels stay.)


def small_weight_grad(g2, x2):
    """dW = dY^T X below the sliced path (the decoder's 8,800 rows, the depth tokens' 30,720): torch.mm with the shipped TunableOp
    choices.  (Round 5 A/B: the same product through the GEMM shim -- hipBLASLt's 32 best heuristic candidates timed per shape, split-K
    ones included -- is 0.83 ms per step SLOWER; the tuned choices already are what the library has.)"""
    return torch.mm(g2.t(), x2)


# (Round 5, measured and NOT used: dW and db of a small linear on a SIDE STREAM beside dX -- the three products are independent and each
# fills a fraction of the chip (138 / 12 / 35 workgroups at 8,800 rows) -- with the side stream waiting for the main one and the main one
# for the side stream inside the node, so that every returned gradient is complete in stream order: +1.55 ms per step (+2.35 in blocks of
# four un-synchronised steps), tools/ab_step.py.  Two cross-stream dependencies per node cost more than the overlap returns on this
# runtime, as round 3 had found for the MSDA plan.)


class _TokenLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        # (kept lean: ~45 of these nodes run in the decoder's backward right behind the matcher's synchronisation, where the GPU
        # waits for the host -- every microsecond of Python here is step time)
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        g2 = grad_out.reshape(-1, grad_out.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = torch.mm(g2, weight).view(x.shape) if need_x else None
        gw = None
        need_b = ctx.has_bias and need_b
        if need_w:
            x2 = x.reshape(-1, x.shape[-1])
            if x2.shape[0] >= MIN_TOKENS:
                gw = weight_grad(g2, x2)
            elif SMALL_WGRAD_KERNEL and linear_wgrad_applies(g2, x2):
                gw, gb = linear_wgrad(g2, x2, need_b, True)       # one pass over dY for both gradients
                return gx, gw, gb
            else:
                gw = small_weight_grad(g2, x2)
        gb = colsum(g2) if need_b else None
        return gx, gw, gb


FAST_LINEAR = True        # `linear()` below: our backward for every GEMM of >= FAST_MIN_ROWS tokens
FAST_MIN_ROWS = 2048


def linear(x, weight, bias=None):
    """``F.linear`` whose backward is ours on the GPU: the bias gradient through the column-sum kernels (10 us for [8800, 256];
    the reduction inside AddmmBackward takes 25), the weight gradient split over K when there are enough tokens."""
    if FAST_LINEAR and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and weight.requires_grad \
            and x.numel() // x.shape[-1] >= FAST_MIN_ROWS:
        return _TokenLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def token_linear(x, linear):
    """``linear(x)`` for an ``nn.Linear``.  On the GPU under autograd the backward is ours: split-K weight gradient
    when there are enough tokens, bias gradient through the column-sum kernel."""
    if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
        # the [L, B, C] view of a batch-major buffer: the GEMM runs on the buffer as it lies (a non-contiguous input costs
        # F.linear a copy and an unfused bias add), the result is handed back as the same kind of view
        return token_linear(x.transpose(0, 1), linear).transpose(0, 1)
    rows = x.numel() // x.shape[-1]
    if x.is_cuda and torch.is_grad_enabled() and linear.weight.requires_grad and \
            (rows >= MIN_ROWS or (FAST_LINEAR and rows >= FAST_MIN_ROWS and x.dtype == torch.float32)):
        return _TokenLinear.apply(x, linear.weight, linear.bias)
    return linear(x)
