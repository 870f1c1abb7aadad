"""Micro-timings of the pointwise library against the PyTorch formulations (one GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd.pointwise import colsum, dropout_add_layernorm, group_norm   # noqa: E402


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
    return a.elapsed_time(b) / n * 1e3


def main():
    for rows, C in [(8800, 256), (30720, 256), (163200, 256), (163200, 128), (163200, 1024)]:
        g = torch.randn(rows, C, device="cuda")
        print("colsum [%d,%d]: ours %.1f us, torch %.1f us" % (rows, C, timeit(lambda: colsum(g)), timeit(lambda: g.sum(0))), flush=True)
    for rows in (8800, 30720, 163200):
        norm = torch.nn.LayerNorm(256).cuda()
        drop = torch.nn.Dropout(0.1).cuda()
        x = torch.randn(rows, 256, device="cuda", requires_grad=True)
        z = torch.randn(rows, 256, device="cuda", requires_grad=True)
        go = torch.randn(rows, 256, device="cuda")

        def ours():
            dropout_add_layernorm(x, z, norm, drop).backward(go)

        def ref():
            norm(x + drop(z)).backward(go)
        print("dropout+add+LN fwd+bwd rows %d: ours %.1f us, torch %.1f us" % (rows, timeit(ours), timeit(ref)), flush=True)
    for shape in [(16, 256, 48, 160), (16, 256, 24, 80), (16, 256, 12, 40)]:
        gn = torch.nn.GroupNorm(32, 256).cuda()
        x = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        go = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
        print("groupnorm+relu fwd+bwd %s: ours %.1f us, torch %.1f us" % (
            shape, timeit(lambda: group_norm(x, gn, relu=True).backward(go)), timeit(lambda: torch.relu(gn(x)).backward(go))), flush=True)


if __name__ == "__main__":
    main()
