"""Dense-FLOP bookkeeping for bench.py's ``roofline.step``.

The library GEMMs / convolutions of a step are counted at the dispatcher (``torch.utils.flop_counter.FlopCounterMode`` sees
aten's mm / addmm / bmm / convolution / convolution_backward, forward and backward, with the gradients that are actually
formed).  The hand-written DENSE kernels of this repo never pass through aten: when counting is on they report their
multiply-adds here (2 flop each).  Off by default: one ``is None`` test per call."""

_acc = None


def start():
    global _acc
    _acc = {}


def stop():
    global _acc
    out, _acc = _acc or {}, None
    return out


def add(kind, flops):
    if _acc is not None:
        _acc[kind] = _acc.get(kind, 0) + int(flops)


def attention_forward(B, H, Lq, Lk, d=32):
    """S = Q K^T and O = P V: 2 products of Lq x Lk x d"""
    add("attention_fwd", 2 * 2 * B * H * Lq * Lk * d)


def attention_backward(B, H, Lq, Lk, d=32):
    """dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q: the 4 products every backward needs, plus the recomputed S = Q K^T
    counted ONCE (bwd_dq and bwd_dkdv each recompute S and dP -- the second evaluation is this implementation's choice, not work
    the operator requires, and is not counted)."""
    add("attention_bwd", 5 * 2 * B * H * Lq * Lk * d)


def conv1x1(M, K, N):
    add("conv1x1_fused", 2 * M * K * N)


def linear_wgrad(R, M, N):
    """dW[M, N] = dY[R, M]^T X[R, N] by csrc/small_wgrad.hip (the product torch.mm would have reported; the column sums are not counted)"""
    add("linear_wgrad", 2 * R * M * N)
