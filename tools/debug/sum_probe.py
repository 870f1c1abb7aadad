"""column-sum variants on decoder-sized matrices: at::sum, gemv against ones, this repo's colsum"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd.pointwise import colsum

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for R, C in ((8800, 256), (8800, 515), (8800, 384), (30720, 256), (30720, 512), (8800, 24), (163200, 256)):
    g = torch.randn(R, C, device="cuda")
    ones = torch.ones(R, device="cuda")
    a = t(lambda: g.sum(0))
    b = t(lambda: torch.mv(g.t(), ones))
    c = t(lambda: colsum(g)) if C % 4 == 0 and C <= 512 else float("nan")
    d = t(lambda: (ones[None, :] @ g))
    err = (torch.mv(g.t(), ones) - g.sum(0)).abs().max().item()
    print("[%6d, %3d]  sum %6.1f us   mv %6.1f us   colsum %6.1f us   ones@g %6.1f us   (mv err %.1e)" % (R, C, a, b, c, d, err))
