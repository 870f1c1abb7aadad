"""Correctness + timing of the fp32 attention kernels against PyTorch (fp64 math reference, SDPA timing)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd import flash_attn as FA   # noqa: E402


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


def main():
    torch.manual_seed(0)
    for B, H, Lq, Lk in [(2, 8, 100, 200), (16, 8, 1920, 1920), (16, 8, 550, 1920)]:
        # nn.MultiheadAttention layout: [L, B, H*32] projections viewed as [B, H, L, 32]
        q = torch.randn(Lq, B, H * 32, device="cuda").view(Lq, B, H, 32).permute(1, 2, 0, 3)
        k = torch.randn(Lk, B, H * 32, device="cuda").view(Lk, B, H, 32).permute(1, 2, 0, 3)
        v = torch.randn(Lk, B, H * 32, device="cuda").view(Lk, B, H, 32).permute(1, 2, 0, 3)
        scale = 1 / math.sqrt(32)
        o, lse = FA.forward(q, k, v, scale, 0.0, 0)
        qd, kd, vd = q.double(), k.double(), v.double()
        s = (qd @ kd.transpose(-1, -2)) * scale
        ref = torch.softmax(s, -1) @ vd
        lse_ref = torch.logsumexp(s, -1) / math.log(2)
        print("B%d H%d Lq%d Lk%d: fwd max err %.2e (ref max %.2f), lse err %.2e" % (
            B, H, Lq, Lk, (o.double() - ref).abs().max().item(), ref.abs().max().item(),
            (lse.view(B, H, Lq).double() - lse_ref).abs().max().item()), flush=True)
        flops = 4.0 * B * H * Lq * Lk * 32
        t = timeit(lambda: FA.forward(q, k, v, scale, 0.0, 0))
        td = timeit(lambda: FA.forward(q, k, v, scale, 0.1, 1234))
        qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
        tt = timeit(lambda: F.scaled_dot_product_attention(qc, kc, vc))
        print("   fwd ours %.3f ms (%.1f TF), with dropout %.3f ms; torch SDPA %.3f ms (%.1f TF)" % (
            t, flops / t / 1e9, td, tt, flops / tt / 1e9), flush=True)
        # backward vs fp64 autograd
        qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        go = torch.randn(Lq, B, H * 32, device="cuda").view(Lq, B, H, 32).permute(1, 2, 0, 3)
        FA.attention(qg, kg, vg).backward(go)
        qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
        (torch.softmax((qd @ kd.transpose(-1, -2)) * scale, -1) @ vd).backward(go.double())
        for n, a_, b_ in (("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
            print("   %s max err %.2e (ref max %.2f)" % (n, (a_.double() - b_).abs().max().item(), b_.abs().max().item()), flush=True)
        o_, lse_ = FA.forward(q, k, v, scale, 0.1, 5)
        goc = go.contiguous()
        tb = timeit(lambda: FA.backward(q, k, v, o_, lse_, goc, scale, 0.1, 5))
        tb0 = timeit(lambda: FA.backward(q, k, v, o_, lse_, goc, scale, 0.0, 0))
        qt, kt_, vt = (t.contiguous().requires_grad_(True) for t in (q, k, v))
        ot = F.scaled_dot_product_attention(qt, kt_, vt, dropout_p=0.1)
        gc = go.contiguous()
        tt2 = timeit(lambda: torch.autograd.grad(ot, (qt, kt_, vt), gc, retain_graph=True))
        print("   bwd ours %.3f ms (no dropout %.3f, %.1f TF at 3.5x fwd flops); torch SDPA bwd (dropout) %.3f ms" % (
            tb, tb0, 3.5 * flops / tb0 / 1e9, tt2), flush=True)
        # dropout statistics: mean preserved, kept fraction
        od, _ = FA.forward(q, k, torch.ones_like(v), scale, 0.1, 77)
        print("   dropout: mean of O with V=1: %.4f (expect 1), std over rows %.4f" % (od.mean().item(), od[..., 0].std().item()), flush=True)


def dropout_backward_check():
    """With dropout the kernels must use the same mask forward and backward: recover P_drop from two forwards with
    one-hot V, then compare the kernel gradients with fp64 autograd through that explicit mask."""
    torch.manual_seed(1)
    B, H, Lq, Lk, p, seed = 2, 3, 70, 64, 0.3, 99
    q = torch.randn(Lq, B, H * 32, device="cuda").view(Lq, B, H, 32).permute(1, 2, 0, 3)
    k = torch.randn(Lk, B, H * 32, device="cuda").view(Lk, B, H, 32).permute(1, 2, 0, 3)
    v = torch.randn(Lk, B, H * 32, device="cuda").view(Lk, B, H, 32).permute(1, 2, 0, 3)
    scale = 1 / math.sqrt(32)
    eye = torch.eye(64, device="cuda")
    pd = torch.cat([FA.forward(q, k, eye[:, i * 32:(i + 1) * 32].expand(B, H, 64, 32).contiguous(), scale, p, seed)[0] for i in range(2)], -1)
    mask = (pd != 0).double()
    print("dropout check: kept fraction %.3f (expect %.3f)" % (mask.mean().item(), 1 - p))
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    pr = torch.softmax((qd @ kd.transpose(-1, -2)) * scale, -1)
    keep_scale = 65536.0 / (65536.0 - round(p * 65536))
    print("   P_drop vs P*mask*scale max err %.2e" % (pd.double() - pr * mask * keep_scale).abs().max().item())
    go = torch.randn(Lq, B, H * 32, device="cuda").view(Lq, B, H, 32).permute(1, 2, 0, 3)
    ((pr * mask * keep_scale) @ vd).backward(go.double())
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    FA.attention(qg, kg, vg, dropout_p=p, seed=seed).backward(go)
    for n, a_, b_ in (("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        print("   %s max err %.2e (ref max %.2f)" % (n, (a_.double() - b_).abs().max().item(), b_.abs().max().item()), flush=True)


if __name__ == "__main__":
    dropout_backward_check()
    main()
