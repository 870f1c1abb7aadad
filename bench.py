#!/usr/bin/env python
"""Headline benchmark: MonoDETR training img/s on KITTI-shaped 1280x384 synthetic images.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward, set criterion (Hungarian matching
included), backward, AdamW update -- ResNet-50 MonoDETR, per-GPU batch 16, fp32 (the reference's
precision), inputs resident in HBM before the timed region.  N > 1: one process per GPU, DDP over
RCCL, per-GPU work fixed ("weak" scaling), value = all images of all ranks / max-over-ranks time.

One JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      -- the dominant MSDA kernel: algorithmic bytes per launch / average
                   launch duration measured here with HIP events on the launch stream / 8 TB/s
  cpu_baseline  -- the CPU restatement (oracle grid_sample formulation, "port") timed on this host,
                   bounded sample (N=1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MATRIX_PEAK_FLOPS = 157.3e12   # MI355X_MICROARCH.md: FP32 vector / matrix peak (v_mfma_f32_32x32x2_f32: the exact-f32 path the step computes in)


def log(msg):
    """Progress to stderr (keeps a long run visibly alive; stdout carries only the JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


_T0 = time.time()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE configs[1]: 16)")
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--resolution", default="1280x384")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "monodetr.yaml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-batch", type=int, default=2)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--graph", action="store_true", help="infer mode: replay the forward from a captured hipGraph")
    ap.add_argument("--miopen-find", action="store_true", help="cudnn.benchmark=True: MIOpen searches per conv shape (slow start)")
    ap.add_argument("--preheat-seconds", type=float, default=3.0,
                    help="untimed steps run for this long before the W warm-up steps: on a fresh box the first ~2 s of "
                         "sustained load run 5 %% slower (clock / power ramp), which 3 warm-up steps (0.3 s) do not cover")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed GLOBAL batch per optimizer step (SURVEY 8d config 3, second definition): every rank runs "
                         "global_batch / (gpus * batch) micro-steps of --batch images with gradient accumulation, one AdamW "
                         "update per step; the line then says \"scaling\": \"strong\" (total work fixed as N grows).  0 (default): "
                         "per-GPU batch fixed, \"weak\"")
    ap.add_argument("--mixed-cameras", action="store_true",
                    help="config 5: the batch mixes fu in {721.5, 552.6, 2055} (per-sample Canonical Object Space scale)")
    ap.add_argument("--no-inference-leg", action="store_true", help="skip the eval-mode leg of the train line")
    ap.add_argument("--inference-steps", type=int, default=20)
    ap.add_argument("--no-dataloader-leg", action="store_true", help="skip the DataLoader-fed leg of the train line")
    ap.add_argument("--dataloader-workers", type=int, default=4, help="reference: 4 (lib/helpers/dataloader_helper.py:21-34)")
    ap.add_argument("--resample-from", default="",
                    help="WxH of the RAW camera images (config 4, 'reference-resampled' variant: 1408x376): the resident batch holds them "
                         "at that size and every step first resamples them to --resolution on the device, which is what the reference's "
                         "loader does on the host for every image (kitti_dataset.py:37,202-206); the model then runs at --resolution")
    ap.add_argument("--no-step-roofline", action="store_true", help="skip the FLOP-counting extra step (roofline.step)")
    ap.add_argument("--no-offsets-probe", action="store_true", help="skip the trained-offsets MSDA probe (roofline.trained_offsets)")
    ap.add_argument("--no-miopen-db", action="store_true",
                    help="ignore the shipped MIOpen find results (monosowa_amd/miopen_db) and use MIOpen's heuristics")
    return ap.parse_args()


def msda_alg_bytes(kind, dims):
    """Algorithmic bytes of one launch (every input read once, every output written once: SURVEY 8d).  The value term of a
    FORWARD launch is capped at the rows the launch can touch at all (Lq * L * P points x 4 corners per head): a 50-query
    inference launch cannot read more than 3,200 of the 10,200 rows per head, and counting all of them would report more
    than the HBM peak.  (The backward writes every grad_value row whatever Lq is.)"""
    B, S, M, D, L, Lq, P = dims
    if kind == "fwd":
        return 4 * B * (min(S, Lq * L * P * 4) * M * D + Lq * M * L * P * 3 + Lq * M * D)
    return 4 * B * (Lq * M * D + min(S, Lq * L * P * 4) * M * D + Lq * M * L * P * 3 + S * M * D + Lq * M * L * P * 3)


def git_head():
    """Short hash of the checked-out commit (+ "-dirty"), or None outside a git checkout (the GPU box's snapshot has no .git:
    tools/collect_pmc.sh and the build step leave it in monosowa_amd/lib/BUILD_COMMIT)."""
    import subprocess
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if h.returncode == 0:
            d = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10)
            return h.stdout.strip() + ("-dirty" if d.stdout.strip() else "")
    except Exception:
        pass
    f = os.path.join(ROOT, "monosowa_amd", "lib", "BUILD_COMMIT")
    return open(f).read().strip() if os.path.exists(f) else None


def dims_S(timer):
    for _, dims, _, _ in timer.records:
        return dims[1]
    return 0


def build_everything(args, device):
    import torch
    import yaml
    from monosowa_amd.helpers.model_helper import build_model
    from monosowa_amd.helpers.optimizer_helper import build_optimizer
    cfg = yaml.safe_load(open(args.config))
    mcfg = cfg["model"]
    mcfg["backbone"] = args.backbone
    mcfg["pretrained"] = False
    mcfg["device"] = device.type
    W, H = (int(x) for x in args.resolution.split("x"))
    mcfg["depth_map_size"] = (W // 16, H // 16)
    torch.manual_seed(cfg.get("random_seed", 444))
    model, criterion = build_model(mcfg)
    model.to(device)
    if device.type == "cuda":
        from monosowa_amd.helpers.model_helper import to_mi355x_layout
        to_mi355x_layout(model)
    criterion.to(device)
    optimizer = build_optimizer(cfg["optimizer"], model)
    return cfg, model, criterion, optimizer, (W, H)


_PROBE_SLEEP = float(os.environ.get("MONOSOWA_PROBE_SLEEP_MS", "0")) * 1e-3     # host-slack probe (tools only)


def train_step_fn(model, criterion, optimizer, accum=1):
    """One optimizer step.  accum > 1: that many micro-steps over the same resident batch with gradient accumulation (each
    micro-batch's losses scaled by 1 / accum; under DDP the all-reduce runs once, with the last micro-step) -- the
    fixed-global-batch definition of a step.  The criterion normalises every micro-batch by its own box count, as the reference
    loop would when it accumulates (trainer_helper.py:123-150 has no accumulation of its own)."""
    import contextlib
    from monosowa_amd.monodetr.criterion import weighted_total
    from monosowa_amd.synthetic import prepare_targets

    def step(batch):
        inputs, calibs, targets, info = batch
        optimizer.zero_grad(set_to_none=True)
        total = None
        for i in range(accum):
            tl = prepare_targets(targets, inputs.shape[0])
            sync_now = i == accum - 1 or not hasattr(model, "no_sync")
            with (contextlib.nullcontext() if sync_now else model.no_sync()):
                outputs = model(inputs, calibs, tl, targets["img_size"])
                loss_dict = criterion(outputs, tl)
                total = weighted_total(loss_dict, criterion.weight_dict)
                if _PROBE_SLEEP:
                    time.sleep(_PROBE_SLEEP)
                (total if accum == 1 else total / accum).backward()
        optimizer.step()
        return total
    return step


def dataloader_leg(args, step, device, world, rank, resolution):
    import torch
    from monosowa_amd.helpers.dataloader_helper import build_dataloader
    from monosowa_amd.helpers.trainer_helper import stage_batch
    n_warm = 2                                           # worker start-up + first batches: untimed
    dcfg = {"type": "synthetic", "batch_size": args.batch, "train_split": "train", "test_split": "val", "resolution": resolution,
            # one dataset for the whole job: under torch.distributed build_dataloader shards it (DistributedSampler), every rank
            # draws steps + n_warm batches of its own images
            "num_samples": args.batch * (args.steps + n_warm) * world}
    loader, _ = build_dataloader(dcfg, workers=args.dataloader_workers, drop_last=True, test=False)
    it = iter(loader)
    for _ in range(n_warm):
        step(stage_batch(next(it), device))
    barrier_sync(device)
    t0 = time.perf_counter()
    n = 0
    for raw in it:
        step(stage_batch(raw, device))
        n += 1
    barrier_sync(device)
    dt = max_over_ranks(time.perf_counter() - t0, device)
    assert n == args.steps, (n, args.steps)
    del it, loader
    return {"value": args.batch * world * n / dt, "unit": "img/s", "steps": n, "ms_per_step": dt / n * 1e3,
            "workers": args.dataloader_workers, "pin_memory": True,
            "workload": "the same train step fed by torch DataLoader(SyntheticKITTI): %d workers, default collate, pinned buffers, "
                        "non_blocking H2D of every key (the reference loop: dataloader_helper.py:21-34, trainer_helper.py:121-127)"
                        % args.dataloader_workers}


def cpu_baseline(args):
    """The CPU path (our restatement; MSDA core = the oracle's grid_sample port of the reference's only
    CPU-capable definition, ms_deform_attn_func.py:41-61) on this host's cores, bounded sample.  BASELINE.md section 2's
    protocol: training at B = 2 (>= 3 timed steps after 1 warm-up) and at B = 16, inference (eval, 50 queries), and the share of
    the step's wall time spent inside the MSDA core."""
    import torch
    import monosowa_amd.ms_deform_attn_func as F
    from oracle import msda_oracle as O
    from monosowa_amd.synthetic import make_batch

    core = {"fwd": 0.0, "bwd": 0.0, "t_bwd": None}

    class _CPUFn:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            # wall time inside the core: the forward directly; the backward between the autograd engine reaching the core's output
            # gradient and the last of its three input gradients (the core's nodes run back to back on the engine's one CPU thread)
            t0 = time.perf_counter()
            out = O.msda_core_torch(value, shapes, loc, w)
            core["fwd"] += time.perf_counter() - t0
            if out.requires_grad:
                def enter(g):
                    core["t_bwd"] = time.perf_counter()
                    core["left"] = sum(1 for t in (value, loc, w) if t.requires_grad)
                    return g

                def leave(g):
                    core["left"] -= 1
                    if core["left"] == 0 and core["t_bwd"] is not None:
                        core["bwd"] += time.perf_counter() - core["t_bwd"]
                        core["t_bwd"] = None
                    return g
                out.register_hook(enter)
                for t in (value, loc, w):
                    if t.requires_grad:
                        t.register_hook(leave)
            return out

    saved = F.MSDeformAttnFunction
    F.MSDeformAttnFunction = _CPUFn
    try:
        cores = torch.get_num_threads()
        cpu = torch.device("cpu")
        cfg, model, criterion, optimizer, (W, H) = build_everything(args, cpu)
        model.train()
        criterion.train()
        step = train_step_fn(model, criterion, optimizer)
        B = args.cpu_baseline_batch
        batch = make_batch(B, cpu, seed=444, resolution=(W, H))
        tw = time.time()
        step(batch)                          # 1 warm-up step, not timed
        tw = time.time() - tw
        core["fwd"] = core["bwd"] = 0.0
        t0 = time.time()
        n = 0
        while n < 3 or (time.time() - t0 < 10.0 and n < 8):      # >= 3 timed steps
            step(batch)
            n += 1
        dt = time.time() - t0
        out = {"value": B * n / dt, "unit": "img/s", "cores": cores, "kind": "port",
               "sample": "%d timed train steps (fwd+criterion+bwd+AdamW) after 1 warm-up step (%.1f s) at batch %d, %dx%d, "
                         "fp32, %.1f s timed" % (n, tw, B, W, H, dt),
               "msda_share": {"value": (core["fwd"] + core["bwd"]) / dt, "forward": core["fwd"] / dt, "backward": core["bwd"] / dt,
                              "what": "wall time inside the MSDA core (grid_sample formulation, 6 calls per step) / step wall time, "
                                      "batch %d train steps; backward by autograd hooks on the core's output / inputs" % B}}
        log("cpu_baseline: batch %d train %.3f img/s, MSDA core %.0f %% of the step" % (B, out["value"], 100 * out["msda_share"]["value"]))
        # ---- inference (eval, 50 queries), batch B ---------------------------------------------------------------------------
        model.eval()
        with torch.no_grad():
            model(batch[0], batch[1], None, batch[2]["img_size"])          # warm-up
            t0 = time.time()
            ni = 0
            while ni < 3 or (time.time() - t0 < 6.0 and ni < 8):
                model(batch[0], batch[1], None, batch[2]["img_size"])
                ni += 1
            dti = time.time() - t0
        out["inference"] = {"value": B * ni / dti, "unit": "img/s",
                            "sample": "%d eval forwards (%d queries) after 1 warm-up at batch %d, %.1f s timed"
                                      % (ni, int(getattr(model, "num_queries", 50)), B, dti)}
        log("cpu_baseline: batch %d inference %.3f img/s" % (B, out["inference"]["value"]))
        # ---- training at the headline batch: ONE timed step, no warm-up of its own (the libraries are warm from the steps above;
        # the B = 2 warm-up step ran 1.3x a timed one) -- bounded: skipped when the B = 2 rate predicts more than 150 s ---------------
        model.train()
        predicted = args.batch / out["value"]
        if args.batch > B and predicted <= 150.0:
            big = make_batch(args.batch, cpu, seed=445, resolution=(W, H))
            t0 = time.time()
            step(big)
            dtb = time.time() - t0
            out["b%d" % args.batch] = {"value": args.batch / dtb, "unit": "img/s",
                                       "sample": "1 timed train step at batch %d (no warm-up step of its own: 1 + %d steps at batch %d ran "
                                                 "before it), %.1f s" % (args.batch, n, B, dtb)}
            log("cpu_baseline: batch %d train %.3f img/s" % (args.batch, out["b%d" % args.batch]["value"]))
        elif args.batch > B:
            out["b%d" % args.batch] = {"value": None, "sample": "skipped: %.0f s predicted for one step from the batch-%d rate" % (predicted, B)}
        return out
    finally:
        F.MSDeformAttnFunction = saved


def count_dense_flops(step, batch):
    """Dense FLOPs of ONE step, counted where they are issued (not from a formula of the model): every aten mm / addmm / bmm /
    convolution / convolution_backward the step dispatches -- forward and the backward parts that exist (no gradient below layer2:
    convolution_backward is counted with the output mask it is called with) -- plus what this repo's own dense kernels report
    (monosowa_amd/flops.py: HIP attention forward / backward, the frozen bottlenecks' fused 1 x 1 convolutions, the small linears' weight gradients).  MSDA's bilinear
    gather / scatter is not a dense contraction and is not in this number (it has its own HBM roofline above).  One extra,
    untimed step."""
    import torch
    from torch.utils.flop_counter import FlopCounterMode
    from monosowa_amd import flops as own
    own.start()
    try:
        with FlopCounterMode(display=False) as fc:
            step(batch)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        lib = {str(getattr(k, "__name__", k)).replace("aten.", ""): int(v) for k, v in fc.get_flop_counts().get("Global", {}).items()}
    finally:
        mine = own.stop()
    return lib, mine


def step_roofline(step, batch, ms_per_step):
    """``roofline.step``: the step's dense FLOPs / its time / the f32 matrix peak (north_star: throughput "as fraction of the HBM/MFMA
    roofline"; the MSDA object above is the HBM side)."""
    try:
        lib, mine = count_dense_flops(step, batch)
    except Exception as e:                                 # the headline must not depend on a counting mode
        return {"bound": "mfma", "error": "%s: %s" % (type(e).__name__, e)}
    total = sum(lib.values()) + sum(mine.values())
    ach = total / (ms_per_step * 1e-3)
    return {"bound": "mfma", "dense_flops_per_step": total, "library_ops": lib, "own_kernels": mine,
            "achieved": ach / 1e12, "peak": F32_MATRIX_PEAK_FLOPS / 1e12, "unit": "TFLOP/s", "frac": ach / F32_MATRIX_PEAK_FLOPS,
            "what": "aten mm/addmm/bmm/convolution(+backward) counted by torch.utils.flop_counter over one extra step, plus the HIP "
                    "attention and fused 1x1-convolution kernels' own counts; / ms_per_step / 157.3 TFLOP/s (f32 matrix peak)"}


def trained_offsets_probe(batch_size, resolution, device, spec="normal:2", iters=20):
    """``roofline.trained_offsets``: the encoder-shape operator pair on sampling offsets that LEFT the module's initial pattern --
    N(0, 2 px) of isotropic drift at every level, the 'normal:2' row of tools/msda_fused_bench.py --sweep
    (profiles/r0*_msda_offset_sweep.json) -- because the bench's own model is a few optimizer steps old and its offsets still sit
    on that pattern: the line's headline kernel time is the best case.  Same entry points, same batch size, HIP-event timed."""
    import torch
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import msda_fused_bench as FB
    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    FB.set_resolution("%dx%d" % resolution)
    value, shapes, lsi, proj, ref, go = FB.make(batch_size, "enc", spec, device)
    B, S, M, D = value.shape
    _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
    t_f = FB.timeit(lambda: MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref), 5, iters)
    t_b = FB.timeit(lambda: MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go), 5, iters)
    dims = (B, S, M, D, 4, S, 4)
    out = {"offsets": spec + " (N(0, sigma px) isotropic drift around the query's own pixel, every level)",
           "msda_bwd_ms": t_b, "msda_fwd_ms": t_f,
           "bwd_frac": msda_alg_bytes("bwd", dims) / (t_b * 1e-3) / HBM_PEAK_BYTES_PER_S,
           "fwd_frac": msda_alg_bytes("fwd", dims) / (t_f * 1e-3) / HBM_PEAK_BYTES_PER_S,
           "what": "tools/msda_fused_bench.py's operator pair at B = %d, S = Lq = %d, %d launches each, back to back (micro-benchmark, not in the step)"
                   % (B, S, iters)}
    del value, proj, ref, go, loc, attw
    torch.cuda.empty_cache()
    return out


class _StdoutGuard:
    """Third-party libraries (RCCL's version banner, MIOpen) write to fd 1; the contract is ONE JSON line on
    stdout.  Everything goes to stderr until ``release()``."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def release(self):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


TUNED_WORKLOADS = (("resnet50", "1280x384", 16), ("resnet101", "1408x376", 16), ("resnet50", "1920x1280", 4))


def tuned_workload(args):
    """Whether the shipped find-db / GEMM choices were MEASURED on this workload (monosowa_amd/miopen_db: BASELINE configs[1],
    config 4 = ResNet-101 at 1408x376, config 5 = 1920x1280 at batch 4; tools/tune_configs.sh).  The database is offered to every
    workload -- MIOpen's immediate mode falls back to its heuristics for a problem it does not hold -- and the line says
    ``"tuned": false`` when the workload is not one of the three."""
    return (args.backbone, args.resolution, args.batch) in TUNED_WORKLOADS and not args.resample_from


def any_rank_says(flag, device):
    """True on EVERY rank as soon as one rank's ``flag`` is true.  Loops whose length depends on a rank's own clock (the
    pre-heat) end through this: a step is a set of collectives, so a rank that granted itself one step more than its
    partners would wait for them forever."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], device=device, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())


def max_over_ranks(seconds, device):
    """The slowest rank's time, on every rank (the contract's "take the MAX over ranks")."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier_sync(device):
    """Barrier + device synchronisation: brackets every timed region on both sides."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize()


def timed_steps(step, batch, steps, device):
    """EXACTLY `steps` steps between two barrier_sync()s; returns the max-over-ranks seconds."""
    barrier_sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(batch)
    barrier_sync(device)
    return max_over_ranks(time.perf_counter() - t0, device)


def preheat(step, batch, seconds, device, sync):
    """Untimed steps for `seconds` of this rank's clock; every rank leaves the loop after the same number of steps."""
    t1, n = time.time(), 0
    while not any_rank_says(time.time() - t1 >= seconds, device):
        step(batch)
        sync()
        n += 1
    return n, time.time() - t1


_TRACE = os.environ.get("MONOSOWA_BENCH_TRACE") == "1" or os.environ.get("MONOSOWA_BENCH_REHEARSAL") == "1"


def trace(msg):
    """Per-RANK progress line (rehearsals and MONOSOWA_BENCH_TRACE=1 only): which rank is where, when a multi-process run is slow or
    stuck.  Costs nothing on the driver's path."""
    if _TRACE:
        sys.stderr.write("[bench %7.1fs rank %s] %s\n" % (time.time() - _T0, os.environ.get("RANK", "0"), msg))
        sys.stderr.flush()


def launch_ranks(args):
    """``python bench.py --gpus N`` started plainly: start the N ranks as a FRESH child (torch.distributed.run) before this
    process has touched the GPU, relay its stdout (rank 0's JSON line) and exit with its code.  Never an exec of a
    process that has initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if (args.gpus > 1 or os.environ.get("MONOSOWA_BENCH_FORCE_LAUNCH") == "1") and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
    guard = _StdoutGuard()
    # measured MIOpen kernel choices for the default workload (monosowa_amd/miopen_tuning.py); before torch loads MIOpen
    from monosowa_amd import miopen_tuning
    tuned = None
    if not args.no_miopen_db and not args.miopen_find:
        tuned = miopen_tuning.use_shipped_db(int(os.environ.get("RANK", "0")))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the CPU path is only the cpu_baseline leg)")
    # rehearsal of the N > 1 path on a ONE-GPU box (tests, tools): every rank on device 0, collectives over gloo (RCCL refuses two
    # ranks on one device).  Never set by the driver: its ranks get one GPU each and RCCL.
    rehearsal = os.environ.get("MONOSOWA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        if world > 2:
            # measured (gpurun_out/reh_b2_3.err): three ranks finish, at 23.6 s per B = 2 train step against 0.028 s per eval step of the
            # same ranks -- the time goes into gloo's host-staged all-reduce of 150 MB of gradients from processes sharing one box's
            # CPU quota, not into the GPU; four ranks (reh4.err, reh_b2_4.err) did not finish a step in 198 - 397 s.  Nothing about
            # the N > 2 path is learnt that way that the world-size-8 gloo tests of the control path do not cover on the CPU.
            raise SystemExit("MONOSOWA_BENCH_REHEARSAL=1 supports WORLD_SIZE <= 2 (more ranks on one device only measure gloo's "
                             "host-staged all-reduce); the N > 2 control path is covered by tests/test_distributed_gloo.py")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # with the shipped find-db MIOpen's immediate mode (benchmark off) already returns the measured winners: no search,
    # no 25 s start-up; --miopen-find runs a fresh search
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    log("MIOpen: %s" % ("shipped find-db " + tuned if tuned else "search" if args.miopen_find else "heuristics"))
    log("torch %s on %s, world %d" % (torch.__version__, torch.cuda.get_device_name(local_rank), world))

    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    from monosowa_amd.helpers.trainer_helper import wrap_ddp
    from monosowa_amd.synthetic import make_batch

    cfg, model, criterion, optimizer, (W, H) = build_everything(args, device)
    train = args.mode == "train"
    model.train(train)
    criterion.train(train)
    model = wrap_ddp(model, device)
    batch = make_batch(args.batch, device, seed=444 + rank, resolution=(W, H), mixed_cameras=args.mixed_cameras)
    batch = (batch[0].contiguous(memory_format=torch.channels_last),) + batch[1:]
    resample = None
    if args.resample_from:
        # "reference-resampled" variant: the resident batch holds the RAW camera images; every step starts by
        # resampling them to the network resolution (bilinear, like the reference loader's PIL affine transform,
        # kitti_dataset.py:202-206) -- on the device, inside the timed region.  The pseudo-labels are normalised coordinates:
        # unchanged by the resampling.
        Wr, Hr = (int(x) for x in args.resample_from.split("x"))
        g = torch.Generator(device=device).manual_seed(444 + rank)
        raw = torch.randn(args.batch, 3, Hr, Wr, device=device, generator=g)

        def resample(b):
            x = torch.nn.functional.interpolate(b[0], size=(H, W), mode="bilinear", align_corners=False)
            return (x.contiguous(memory_format=torch.channels_last),) + tuple(b[1:])
        batch = (raw,) + batch[1:]
    accum = 1
    if args.global_batch:
        assert train and args.global_batch % (args.batch * world) == 0, "--global-batch must be a multiple of gpus * batch"
        accum = args.global_batch // (args.batch * world)

    if train:
        step0 = train_step_fn(model, criterion, optimizer, accum)
    elif not args.graph:
        def step0(b):
            with torch.no_grad():
                return model(b[0], b[1], None, b[2]["img_size"])["pred_logits"]
    else:
        from monosowa_amd.helpers.tester_helper import GraphedForward
        assert resample is None, "--graph replays a captured forward of the resident batch"
        graphed = GraphedForward(model, batch[0], batch[1], batch[2]["img_size"])

        def step0(b):
            return graphed(b[0], b[1], b[2]["img_size"])["pred_logits"]
    step = step0 if resample is None else (lambda b: step0(resample(b)))

    def sync():
        barrier_sync(device)

    log("model + batch ready; warm-up")
    if args.preheat_seconds > 0:
        trace("first step: enter")
        step(batch)                       # first step: library initialisation, MIOpen kernel selection
        torch.cuda.synchronize()
        trace("first step: done")
        n, dt = preheat(step, batch, args.preheat_seconds, device, torch.cuda.synchronize)
        log("pre-heat: %d untimed steps in %.1f s (not counted as warm-up steps)" % (n, dt))
    for i in range(args.warmup):
        t1 = time.time()
        step(batch)
        torch.cuda.synchronize()
        log("warm-up step %d: %.2f s" % (i, time.time() - t1))
        trace("warm-up step %d done" % i)

    with MSDA.LaunchTimer() as timer:
        elapsed = timed_steps(step, batch, args.steps, device)
    log("timed %d steps: %.3f s" % (args.steps, elapsed))
    trace("timed region done")

    # the device assignment solver reports an invalid / oversized cost matrix through a deferred status word: look at it for certain
    matcher = getattr(criterion, "matcher", None)
    if train and matcher is not None and hasattr(matcher, "check_device_status"):
        matcher.check_device_status(block=True)

    global_batch = args.batch * world * accum
    value = global_batch * args.steps / elapsed

    # ---- roofline.step: the step's dense FLOPs counted over ONE extra (untimed) step; every rank runs it (a step is a set of collectives)
    step_roof = None
    if train and not args.no_step_roofline and accum == 1:
        step_roof = step_roofline(step, batch, elapsed / args.steps * 1e3)
        log("roofline.step: %s" % (("%.1f TFLOP/step, %.1f %% of the f32 matrix peak" % (step_roof["dense_flops_per_step"] / 1e12, 100 * step_roof["frac"]))
                                   if "frac" in step_roof else step_roof.get("error")))

    # ---- DataLoader-fed leg ("separately, with a synthetic DataLoader"): the same K train steps, every batch
    # coming out of build_dataloader(SyntheticKITTI) -- worker processes, collate, pinned host buffers, non-blocking H2D copies --
    # the way the reference loop is fed (lib/helpers/dataloader_helper.py:21-34: 4 workers; trainer_helper.py:121-127: per-key
    # .to(device)).  Timed like the resident leg; `value` above stays the resident number.
    dataloader = None
    # (N = 1 only, like the cpu_baseline leg: at N > 1 the line's job is the scaling curve, and 4 forked workers per rank next to RCCL
    # add nothing to it)
    if train and world == 1 and not args.no_dataloader_leg and resample is None and accum == 1:
        dataloader = dataloader_leg(args, step0, device, world, rank, (W, H))
        log("dataloader leg: %d steps in %.3f s" % (args.steps, dataloader["ms_per_step"] * args.steps * 1e-3))

    # ---- inference leg of the train line (north_star: "training/inference throughput ... reported"): eval-mode forward of
    # the same model on the same resident batch, every rank its own images, timed like the train leg -----------------------
    inference = None
    if train and not args.no_inference_leg:
        net = model.module if hasattr(model, "module") else model
        net.eval()

        def infer(b):
            with torch.no_grad():
                return net(b[0], b[1], None, b[2]["img_size"])["pred_logits"]
        ibatch = batch if resample is None else resample(batch)
        for _ in range(3):
            infer(ibatch)
        with MSDA.LaunchTimer() as itimer:
            i_elapsed = timed_steps(infer, ibatch, args.inference_steps, device)
        eval_queries = int(getattr(net, "num_queries", 50))
        enc = [(dims, d) for (kind, dims), d in itimer.summary().items() if kind == "fwd" and dims[5] == dims[1]]
        inference = {"value": args.batch * world * args.inference_steps / i_elapsed, "unit": "img/s", "steps": args.inference_steps,
                     "ms_per_step": i_elapsed / args.inference_steps * 1e3, "queries": eval_queries,
                     "workload": "eval forward (%d queries), per-GPU batch %d" % (eval_queries, args.batch)}
        if enc:
            dims, d = enc[0]
            ab = msda_alg_bytes("fwd", dims)
            inference["msda_fwd"] = {"Lq": dims[5], "avg_launch_ms": d["avg_ms"], "alg_bytes_per_launch": ab,
                                     "frac": ab / (d["avg_ms"] * 1e-3) / HBM_PEAK_BYTES_PER_S}
        net.train()
        log("inference leg: %d eval steps in %.3f s" % (args.inference_steps, i_elapsed))

    kernels = []
    for (kind, dims), d in sorted(timer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        ab = msda_alg_bytes(kind, dims)
        ach = ab / (d["avg_ms"] * 1e-3)
        kernels.append({"kernel": "msda_%s" % kind, "Lq": dims[5], "B": dims[0], "launches_per_step": d["launches"] / args.steps,
                        "avg_ms": d["avg_ms"], "alg_bytes": ab, "achieved_GBps": ach / 1e9, "frac": ach / HBM_PEAK_BYTES_PER_S})
    roofline = None
    if kernels:
        k = kernels[0]
        # HBM traffic per launch from the PMC counters: collected by tools/collect_pmc.sh in separate rocprofv3 --pmc passes on
        # the kernels this step runs; the file names its source profile and the kernel templates it was read from
        traffic = traffic_src = traffic_at = None
        tf = os.path.join(ROOT, "profiles", "msda_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get("%s_Lq%d_B%d" % (k["kernel"], k["Lq"], k["B"]))
                traffic_src = tj.get("_source")
                traffic_at = tj.get("_collected_at")
            except Exception:
                traffic = None
        tot_bytes = sum(x["alg_bytes"] * x["launches_per_step"] for x in kernels)
        tot_ms = sum(x["avg_ms"] * x["launches_per_step"] for x in kernels)
        fwd = next((x for x in kernels if x["kernel"] == "msda_fwd" and x["Lq"] == k["Lq"]), None)
        roofline = {"bound": "hbm", "kernel": "%s(B=%d,Lq=%d,S=%d,M=8,D=32,L=4,P=4)" % (k["kernel"], k["B"], k["Lq"], dims_S(timer)),
                    "achieved": k["achieved_GBps"], "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                    "frac": k["frac"], "traffic": traffic, "traffic_source": traffic_src,
                    # the counters come from separate rocprofv3 --pmc passes (tools/collect_pmc.sh), not from this run: the commit
                    # whose kernels they were collected on, and the commit this line was measured at
                    "traffic_collected_at": traffic_at, "measured_at": git_head(), "avg_launch_ms": k["avg_ms"],
                    "alg_bytes_per_launch": k["alg_bytes"],
                    "forward_same_shape": None if fwd is None else {"avg_launch_ms": fwd["avg_ms"], "achieved": fwd["achieved_GBps"],
                                                                    "frac": fwd["frac"]},
                    "all_msda_aggregate": {"alg_bytes_per_step": tot_bytes, "ms_per_step": tot_ms,
                                           "achieved": tot_bytes / (tot_ms * 1e-3) / 1e9,
                                           "frac": tot_bytes / (tot_ms * 1e-3) / HBM_PEAK_BYTES_PER_S, "target_frac": 0.60},
                    "all_msda_kernels": kernels}
        if step_roof is not None:
            roofline["step"] = step_roof
        if train and world == 1 and not args.no_offsets_probe and k["Lq"] == dims_S(timer):
            try:
                roofline["trained_offsets"] = trained_offsets_probe(args.batch, (W, H), device)
            except Exception as e:
                roofline["trained_offsets"] = {"error": "%s: %s" % (type(e).__name__, e)}

    net_ = model.module if hasattr(model, "module") else model
    eval_queries_cfg = int(getattr(net_, "num_queries", 50))
    train_queries = eval_queries_cfg * int(getattr(net_, "group_num", 11))
    if rank == 0:
        line = {
            "metric": "MonoDETR %s img/s (KITTI %dx%d)" % ("training" if train else "inference", W, H),
            "value": value, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            # weak: per-GPU batch fixed as N grows; strong: --global-batch fixed (gradient accumulation fills the difference)
            "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MonoDETR %s KITTI %dx%d, %s step, per-GPU batch %d, synthetic images + random pseudo-labels"
                                   % (args.backbone, W, H, ("resample %s -> " % args.resample_from if args.resample_from else "") +
                                      ("fwd+criterion+bwd+AdamW" if train else "eval fwd (%d queries)" % eval_queries_cfg), args.batch),
                       "global_batch": global_batch, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                       "accumulation_micro_steps": accum, "mixed_cameras": bool(args.mixed_cameras),
                       "queries": train_queries if train else eval_queries_cfg, "tuned": tuned_workload(args),
                       "resampled_from": args.resample_from or None},
            "roofline": roofline,
        }
        if inference is not None:
            line["inference"] = inference
        if dataloader is not None:
            line["dataloader"] = dataloader
        if dist.is_initialized():          # the CPU leg below must not meet an RCCL-only process group
            dist.barrier()
            dist.destroy_process_group()
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            log("cpu_baseline leg (bounded sample)")
            line["cpu_baseline"] = cpu_baseline(args)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        guard.release()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
