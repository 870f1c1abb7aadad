"""Kernel micro-benchmark for the MSDA HIP kernels (SURVEY.md section 8d "kernel micro-benchmark
inputs"): value ~ N(0,1); loc = pixel-centre reference grid + U(-4,4) px (encoder) or U(0,1)
(decoder); w = softmax(N(0,1)) over 16; B=16, 1280x384 levels; 20 warm-up + 100 timed launches.
Prints achieved algorithmic GB/s against the 8 TB/s HBM peak."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

LEVELS = [(48, 160), (24, 80), (12, 40), (6, 20)]
INIT_LIKE = os.environ.get("MSDA_BENCH_INIT_LIKE") == "1"     # encoder offsets as the freshly initialised module produces them
HBM_PEAK = 8.0e12


def make(B, Lq_kind, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    shapes = torch.tensor(LEVELS, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    starts = [0]
    for h, w in LEVELS[:-1]:
        starts.append(starts[-1] + h * w)
    MSDA.attach_host_geometry(shapes, lsi, LEVELS, starts)     # what the transformer does (no sync in forward)
    S, M, D, L, P = 10200, 8, 32, 4, 4
    value = torch.randn(B, S, M, D, device=dev, generator=g)
    if Lq_kind == "enc":
        Lq = S
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                         for h, w in LEVELS])
        if INIT_LIKE:
            # the module's initial sampling_offsets bias (ms_deform_attn.py:106-114): head m looks along angle 2 pi m / M,
            # point p sits p + 1 pixels out (direction normalised to max-abs 1) + a little learned jitter
            import math
            th = torch.arange(M, dtype=torch.float32, device=dev) * (2.0 * math.pi / M)
            d = torch.stack([th.cos(), th.sin()], -1)
            d = d / d.abs().max(-1, keepdim=True)[0]
            px = d[None, None, :, None, None, :] * torch.arange(1, P + 1, device=dev)[None, None, None, None, :, None]
            px = px + 0.02 * torch.randn(B, Lq, M, L, P, 2, device=dev, generator=g)
            off = px / shapes.flip(1)[None, None, None, :, None, :]
        else:
            off = (torch.rand(B, Lq, M, L, P, 2, device=dev, generator=g) * 8 - 4) / shapes.flip(1)[None, None, None, :, None, :]
        loc = (ref[None, :, None, None, None, :] + off).contiguous()
    else:
        Lq = int(Lq_kind)
        loc = torch.rand(B, Lq, M, L, P, 2, device=dev, generator=g)
    w = torch.softmax(torch.randn(B, Lq, M, L * P, device=dev, generator=g), -1).view(B, Lq, M, L, P).contiguous()
    go = torch.randn(B, Lq, M * D, device=dev, generator=g)
    return value, shapes, lsi, loc, w, go


def alg_bytes(B, S, M, D, L, P, Lq, bwd):
    fwd = 4 * (S * M * D + Lq * M * L * P * 3 + Lq * M * D)
    if not bwd:
        return B * fwd
    # reads grad_out + value + loc + w, writes grad_value + grad_loc + grad_w  
    return B * 4 * (Lq * M * D + S * M * D + Lq * M * L * P * 3 + S * M * D + Lq * M * L * P * 3)


def timeit(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kinds", default="enc,550,50")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for kind in a.kinds.split(","):
        value, shapes, lsi, loc, w, go = make(a.batch, kind, dev)
        Lq = loc.shape[1]
        tf = timeit(lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, w, 64), a.warmup, a.iters)
        tb = timeit(lambda: MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, w, go, 64), a.warmup, a.iters)
        for name, t, bwd in (("fwd", tf, False), ("bwd", tb, True)):
            ab = alg_bytes(a.batch, 10200, 8, 32, 4, 4, Lq, bwd)
            rows.append({"kernel": name, "Lq": Lq, "B": a.batch, "ms": t * 1e3, "alg_bytes": ab,
                         "GBps": ab / t / 1e9, "frac_hbm_peak": ab / t / HBM_PEAK})
            print(json.dumps(rows[-1]), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
