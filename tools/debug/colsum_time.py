"""colsum() at the decoder's bias-gradient shapes (MONOSOWA_POINTWISE_LIB selects another build, e.g. -DMONO_COLSUM_STRIP_ROWS=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd.pointwise import colsum


def timeit(fn, n=200):
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
for rows, C in [(8800, 256), (8800, 384), (8800, 512), (2200, 256), (16384, 256), (30720, 256)]:
    g = torch.randn(rows, C, device="cuda")
    err = (colsum(g) - g.double().sum(0).float()).abs().max().item()
    out.append("[%d,%d] %.1f us (err %.1e)" % (rows, C, timeit(lambda: colsum(g)), err))
print(" | ".join(out))
