"""dropout_add_layernorm forward / backward kernels at the encoder's and the decoder's row counts (MONOSOWA_POINTWISE_LIB selects another
build, e.g. -DMONO_LN_ROWS=1): time per call by events, bytes moved / time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd import pointwise as PW


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = []
w, bb = torch.randn(256, device="cuda"), torch.randn(256, device="cuda")
for rows in (163200, 30720, 8800):
    x, z, gy = (torch.randn(rows, 256, device="cuda") for _ in range(3))
    y, s, mean, rstd, seed = PW.ln_forward(x, z, w, bb, 0.1, 1e-5)
    tf = timeit(lambda: PW.ln_forward(x, z, w, bb, 0.1, 1e-5))
    tb = timeit(lambda: PW.ln_backward(gy, s, mean, rstd, w, 0.1, seed, with_gz_sum=True))
    nbytes = rows * 256 * 4 * 4
    out.append("rows %d: fwd %.1f us (%.2f TB/s), bwd %.1f us (%.2f TB/s)" % (rows, tf, nbytes / tf / 1e6, tb, nbytes / tb / 1e6))
print(" | ".join(out))
