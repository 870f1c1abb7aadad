"""Times the attention kernels at the two shapes of the train step (dropout 0.1): depth encoder 1920 x 1920, decoder depth
cross-attention 550 x 1920; B = 16, H = 8.  MONOSOWA_ATTN_LIB selects another build (tools/debug/variants/attn_*.so)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd import flash_attn as FA


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = []
for B, H, Lq, Lk in [(16, 8, 1920, 1920), (16, 8, 550, 1920)]:
    mk = lambda L: torch.randn(L, B, H * 32, device="cuda").view(L, B, H, 32).permute(1, 2, 0, 3)
    q, k, v, go = mk(Lq), mk(Lk), mk(Lk), mk(Lq)
    scale = 1 / math.sqrt(32)
    o, lse = FA.forward(q, k, v, scale, 0.1, 5)
    tf = timeit(lambda: FA.forward(q, k, v, scale, 0.1, 5))
    goc = go.contiguous()
    tb = timeit(lambda: FA.backward(q, k, v, o, lse, goc, scale, 0.1, 5))
    flops = 4.0 * B * H * Lq * Lk * 32
    line = "%dx%d: fwd %.3f ms (%.0f TF) bwd %.3f ms (%.0f TF at 3.5x)" % (Lq, Lk, tf, flops / tf / 1e9, tb, 3.5 * flops / tb / 1e9)
    if hasattr(FA, "keep_bits_like"):               # the training path: the forward saves the dropout mask, the backward reads it
        bits = FA.keep_bits_like(q, k, 0.1)
        FA.forward(q, k, v, scale, 0.1, 5, keep_bits=bits)
        tf2 = timeit(lambda: FA.forward(q, k, v, scale, 0.1, 5, keep_bits=bits))
        tb2 = timeit(lambda: FA.backward(q, k, v, o, lse, goc, scale, 0.1, 5, keep_bits=bits))
        line += "; with saved keep bits: fwd %.3f, bwd %.3f ms" % (tf2, tb2)
    out.append(line)
print(" | ".join(out))
