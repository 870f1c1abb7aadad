"""Micro-benchmark of the operator the train step runs: the FUSED MSDA entry points on a merged projection
(msda_fused_forward_strided_f32 / msda_fused_backward_strided_f32; ops/modules/ms_deform_attn.py:145-160 inside the
kernels), B = 16, 1280x384 pyramid.

    python tools/msda_fused_bench.py [--kinds enc,550] [--iters 50] [--offsets init|uniform] [--out file.json]

encoder shape: Lq = S = 10200, reference points = pixel centres (depthaware_transformer.py:363-376), sampling offsets
either the module's initial pattern (head m along angle 2 pi m / 8, point p at p + 1 pixels, ms_deform_attn.py:106-114,
plus N(0, 0.3) px of learned drift) or U(-4, 4) px.  decoder shape: Lq = 550 queries, 6-d reference boxes.
Prints per-op time and achieved algorithmic GB/s against the 8 TB/s HBM peak (SURVEY.md 8d byte counts).  This is the
command the PMC traffic of profiles/msda_traffic.json is collected on (tools/collect_pmc.sh)."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

LEVELS = [(48, 160), (24, 80), (12, 40), (6, 20)]          # 1280 x 384; --resolution WxH replaces it (strides 8, 16, 32, 64)
HBM_PEAK = 8.0e12


def set_resolution(res):
    W, H = (int(x) for x in res.split("x"))
    LEVELS[:] = [(-(-H // st), -(-W // st)) for st in (8, 16, 32, 64)]


def make(B, kind, offsets, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    shapes = torch.tensor(LEVELS, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, LEVELS, lsi.tolist())
    S, M, D, L, P = sum(h * w for h, w in LEVELS), 8, 32, 4, 4
    value = torch.randn(B, S, M, D, device=dev, generator=g)
    if kind == "enc":
        Lq = S
        centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                                        indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in LEVELS])
        ref = centres[None, :, None, :].expand(B, Lq, L, 2).contiguous()
    else:
        Lq = int(kind)
        ref = torch.cat([torch.rand(B, Lq, 1, 2, device=dev, generator=g).expand(B, Lq, L, 2),
                         torch.rand(B, Lq, 1, 4, device=dev, generator=g).expand(B, Lq, L, 4) * 0.2], -1).contiguous()
    if offsets.startswith("normal:"):
        # isotropic learned drift of sigma pixels (of each sampled level) around the query's own pixel
        off = float(offsets.split(":")[1]) * torch.randn(B, Lq, M, L, P, 2, device=dev, generator=g)
    elif offsets == "init" or offsets.startswith("init:"):
        # init:<sigma px> -- the drift around the pattern; 0.05 is what the train step bench.py times shows after its warm-up
        # (tools/debug/offset_stats.py: 0.02 - 0.09 px per (head, level, point)), the default 0.3 a few hundred steps later
        drift = float(offsets.split(":")[1]) if ":" in offsets else 0.3
        th = torch.arange(M, dtype=torch.float32, device=dev) * (2.0 * math.pi / M)
        d = torch.stack([th.cos(), th.sin()], -1)
        d = d / d.abs().max(-1, keepdim=True)[0]
        off = d[None, None, :, None, None, :] * torch.arange(1, P + 1, device=dev)[None, None, None, None, :, None]
        off = off + drift * torch.randn(B, Lq, M, L, P, 2, device=dev, generator=g)
    else:
        off = torch.rand(B, Lq, M, L, P, 2, device=dev, generator=g) * 8 - 4
    logits = torch.randn(B, Lq, M, L * P, device=dev, generator=g)
    proj = torch.cat([off.reshape(B, Lq, M * 32), logits.reshape(B, Lq, M * 16)], -1).contiguous()
    go = torch.randn(B, Lq, M * D, device=dev, generator=g)
    return value, shapes, lsi, proj, ref, go


def trained_projections(steps, dev):
    """The [B, S, 384] sampling-offset | attention-logit projections of the three encoder layers after `steps` optimizer steps
    on one synthetic batch (the loop of tools/overfit_check.py), captured at the fused operator's entry."""
    import yaml
    from monosowa_amd import miopen_tuning
    miopen_tuning.use_shipped_db(0)
    from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
    from monosowa_amd.helpers.optimizer_helper import build_optimizer
    from monosowa_amd.monodetr.criterion import weighted_total
    from monosowa_amd.synthetic import make_batch, prepare_targets
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "monodetr.yaml")))
    torch.manual_seed(444)
    model, crit = build_model(cfg["model"])
    model = to_mi355x_layout(model.to(dev)).train()
    crit.to(dev).train()
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, _ = make_batch(16, dev)
    inputs = inputs.contiguous(memory_format=torch.channels_last)
    for i in range(steps):
        tl = prepare_targets(targets, 16)
        opt.zero_grad(set_to_none=True)
        weighted_total(crit(model(inputs, calibs, tl, targets["img_size"]), tl), crit.weight_dict).backward()
        opt.step()
    captured = []
    real = MSDA.ms_deform_attn_fused_forward_merged_save

    def spy(value, shapes, lsi, proj, ref, value_mask=None):
        if proj.shape[1] == value.shape[1]:
            captured.append((value.detach().clone(), proj.detach().clone(), ref.detach().clone()))
        return real(value, shapes, lsi, proj, ref, value_mask)
    MSDA.ms_deform_attn_fused_forward_merged_save = spy
    try:
        tl = prepare_targets(targets, 16)
        model(inputs, calibs, tl, targets["img_size"])
    finally:
        MSDA.ms_deform_attn_fused_forward_merged_save = real
    torch.cuda.synchronize()
    del model, crit, opt
    torch.cuda.empty_cache()
    return captured


def offset_statistics(proj, ref, M=8):
    """What the kernels' plan sees: d = floor(loc * size - 0.5) - centre floor of the query's own pixel, per axis, over all
    points: |d| quantiles and the share of points beyond the isotropic window (halo 5) / the scan capacity (8 px)."""
    B, Lq, _ = proj.shape
    off = proj[:, :, :M * 32].reshape(B, Lq, M, 4, 4, 2)
    sizes = torch.tensor([[w, h] for h, w in LEVELS], dtype=torch.float32, device=proj.device)            # (W, H) per level
    loc = ref[:, :, None, :, None, :] + off / sizes[None, None, None, :, None, :]
    low = torch.floor(loc * sizes[None, None, None, :, None, :] - 0.5)
    cen = torch.floor(ref[:, :, None, :, None, :] * sizes[None, None, None, :, None, :] - 0.5)
    d = (low - cen).abs().amax(-1).flatten()
    q = torch.quantile(d[:: max(1, d.numel() // 2000000)].float(), torch.tensor([0.5, 0.9, 0.99], device=d.device))
    return {"abs_d_median_px": q[0].item(), "abs_d_p90_px": q[1].item(), "abs_d_p99_px": q[2].item(),
            "beyond_window_halo5": (d > 5).float().mean().item(), "beyond_scan_reach8": (d > 8).float().mean().item()}


def alg_bytes(B, S, M, D, L, P, Lq, bwd):
    fwd = 4 * (S * M * D + Lq * M * L * P * 3 + Lq * M * D)
    if not bwd:
        return B * fwd
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
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kinds", default="enc,550")
    ap.add_argument("--offsets", default="init",
                    help="init (the module's initial pattern + N(0, 0.3) px), uniform (U(-4, 4) px), normal:<sigma px>, "
                         "trained[:steps] (the encoder layers' projections of a model trained for that many steps on one synthetic "
                         "batch, tools/overfit_check.py's loop; default 150)")
    ap.add_argument("--sweep", default=None, help="comma-separated --offsets values: one table / JSON over all of them (encoder shape)")
    ap.add_argument("--resolution", default=None, help="WxH of the image (default 1280x384): config 4 = 1408x376, config 5 = 1920x1280")
    ap.add_argument("--recompute", action="store_true", help="ABI v5 pair (backward re-evaluates the prologue)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {}
    if a.resolution:
        set_resolution(a.resolution)
    if a.sweep:
        # one row per offset distribution, encoder shape: forward / backward time and what the plan sees
        for spec in a.sweep.split(","):
            cases = []
            if spec.startswith("trained"):
                steps = int(spec.split(":")[1]) if ":" in spec else 150
                value, shapes, lsi, _, _, go = make(a.batch, "enc", "init", dev)
                for i, (v, proj, ref) in enumerate(trained_projections(steps, dev)):
                    cases.append(("%s/layer%d" % (spec, i), v.view(a.batch, -1, 8, 32).contiguous(), proj.contiguous(), ref.contiguous()))
            else:
                value, shapes, lsi, proj, ref, go = make(a.batch, "enc", spec, dev)
                cases.append((spec, value, proj, ref))
            for name, value, proj, ref in cases:
                _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
                t_f = timeit(lambda: MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref), a.warmup, a.iters)
                t_b = timeit(lambda: MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go), a.warmup, a.iters)
                row = dict(offset_statistics(proj, ref), fwd_ms=t_f, bwd_ms=t_b)
                res[name] = row
                print("%-18s fwd %.3f ms  bwd %.3f ms  |d| p50 %.1f p90 %.1f p99 %.1f px  beyond halo 5: %.1f %%  beyond reach 8: %.1f %%"
                      % (name, t_f, t_b, row["abs_d_median_px"], row["abs_d_p90_px"], row["abs_d_p99_px"],
                         100 * row["beyond_window_halo5"], 100 * row["beyond_scan_reach8"]), flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)
        return
    for kind in a.kinds.split(","):
        value, shapes, lsi, proj, ref, go = make(a.batch, kind, a.offsets, dev)
        B, S, M, D = value.shape
        Lq = proj.shape[1]
        if MSDA.fused_save_supported(value, shapes, lsi, Lq, ref.shape[-1]) and not a.recompute:
            # what the train step runs at the self-attention shape (ABI v6): the forward stores locations / weights
            # level-major, the backward reads them
            _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
            t_f = timeit(lambda: MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref), a.warmup, a.iters)
            t_b = timeit(lambda: MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go), a.warmup, a.iters)
        else:
            t_f = timeit(lambda: MSDA.ms_deform_attn_fused_forward_merged(value, shapes, lsi, proj, ref), a.warmup, a.iters)
            t_b = timeit(lambda: MSDA.ms_deform_attn_fused_backward_merged(value, shapes, lsi, proj, ref, go), a.warmup, a.iters)
        for name, t, bwd in (("fwd", t_f, False), ("bwd", t_b, True)):
            nbytes = alg_bytes(B, S, M, D, 4, 4, Lq, bwd)
            res["msda_%s_Lq%d_B%d" % (name, Lq, B)] = {"ms": t, "alg_bytes": nbytes, "GBps": nbytes / t / 1e6, "frac_of_8TBps": nbytes / (t * 1e-3) / HBM_PEAK}
            print("fused msda_%s Lq=%-6d B=%d offsets=%s: %.3f ms  %.0f GB/s algorithmic = %.1f %% of HBM peak"
                  % (name, Lq, B, a.offsets, t, nbytes / t / 1e6, 100 * nbytes / (t * 1e-3) / HBM_PEAK), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
