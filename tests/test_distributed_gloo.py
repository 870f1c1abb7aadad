"""world_size-2 checks of the data-parallel path on CPU (gloo): the DDP-wrapped model's gradients are
the mean of the ranks' local gradients (bucketed all-reduce, never-used parameters frozen), num_boxes
is normalised over the whole job (monodetr.py:1202-1206), replicas stay identical after the step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed=444):
    import yaml
    import monosowa_amd.ms_deform_attn_func as F
    from oracle import msda_oracle as O
    from monosowa_amd.helpers.model_helper import build_model
    from monosowa_amd.helpers.optimizer_helper import build_optimizer

    class _Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return O.msda_core_torch(value, shapes, loc, w)
    F.MSDeformAttnFunction = _Fn
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "monodetr.yaml")))
    mcfg = dict(cfg["model"], device="cpu", depth_map_size=(12, 4), dropout=0.0)
    torch.manual_seed(seed)
    model, crit = build_model(mcfg)
    opt = build_optimizer(cfg["optimizer"], model)
    return model.train(), crit.train(), opt


def _step(model, crit, opt, inputs, calibs, targets):
    from monosowa_amd.synthetic import prepare_targets
    tl = prepare_targets(targets, inputs.shape[0])
    opt.zero_grad(set_to_none=True)
    out = model(inputs, calibs, tl, targets["img_size"])
    ld = crit(out, tl)
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    total.backward()
    opt.step()
    return total.detach()


def _grads(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from monosowa_amd.helpers.trainer_helper import wrap_ddp
    from monosowa_amd.monodetr import misc
    from monosowa_amd.synthetic import make_batch, prepare_targets
    model, crit, opt = _build()
    ddp = wrap_ddp(model, torch.device("cpu"))
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    assert all(not p.requires_grad for n, p in model.named_parameters() if n in set(model.unused_parameter_names()))
    inputs, calibs, targets, _ = make_batch(2, "cpu", seed=7 + rank, resolution=(192, 64))
    tl = prepare_targets(targets, 2)

    def loss_of(out):
        ld = crit(out, tl)
        return sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict), ld

    # (1) local gradient without synchronisation (the depth predictor keeps the reference's hard-coded
    # dropout 0.1, so both passes are seeded identically)
    torch.manual_seed(100 + rank)
    with ddp.no_sync():
        total, ld = loss_of(ddp(inputs, calibs, tl, targets["img_size"]))
        total.backward()
    local = _grads(model).clone()
    ddp.zero_grad(set_to_none=True)
    # (2) the same step through the bucketed all-reduce: must equal the mean of the local gradients
    torch.manual_seed(100 + rank)
    total, ld = loss_of(ddp(inputs, calibs, tl, targets["img_size"]))
    total.backward()
    synced = _grads(model).clone()
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    mean = (both[0] + both[1]) / world
    assert (synced - mean).abs().max() <= 1e-4 * mean.abs().max(), (synced - mean).abs().max()
    # (3) num_boxes is normalised over the whole job: loss_center = local L1 sum / (all boxes * groups / world)
    n_local = torch.tensor([float(sum(len(t["labels"]) for t in tl))])
    n_all = n_local.clone()
    dist.all_reduce(n_all)
    crit_num_boxes = float(n_all) * crit.group_num / world
    from monosowa_amd.monodetr.misc import is_dist_avail_and_initialized
    assert is_dist_avail_and_initialized()
    with torch.no_grad():
        out = ddp(inputs, calibs, tl, targets["img_size"])
        out_wo_aux = {k: v for k, v in out.items() if k != "aux_outputs"}
        ind = crit.matcher(out_wo_aux, tl, group_num=crit.group_num)
        idx = crit._get_src_permutation_idx(ind)
        tgt = torch.cat([t["boxes_3d"][:, 0:2][i] for t, (_, i) in zip(tl, ind)])
        manual = (out["pred_boxes"][:, :, 0:2][idx] - tgt).abs().sum() / crit_num_boxes
        got = crit(out, tl)["loss_center"]
    assert torch.allclose(got, manual, rtol=1e-5), (got, manual)
    # (4) optimizer step keeps the replicas identical; logging reduction averages over ranks
    opt.step()
    w = model.class_embed[0].weight.detach().clone()
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    assert torch.equal(gathered[0], gathered[1])
    red = misc.reduce_dict({"loss": total.detach()})
    losses = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(losses, total.detach())
    assert torch.allclose(red["loss"], (losses[0] + losses[1]) / 2)
    if rank == 0:
        q.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_gradient_allreduce_and_num_boxes():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get() == "ok"


def test_bench_self_launch_spawns_a_child_and_returns_its_code():
    """``python bench.py --gpus N`` started plainly (no RANK in the environment) must start its ranks as a FRESH child
    (torch.distributed.run) before touching the GPU and exit with the child's code (bench.py: launch_ranks).  Here there is
    no GPU: the single rank stops with bench.py's own "needs an MI355X" message -- what matters is that it is the CHILD that
    says so, under a launcher, and that the parent relays the failure code instead of running the benchmark itself."""
    import subprocess
    import sys
    env = dict(os.environ, MONOSOWA_BENCH_FORCE_LAUNCH="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert "starting 1 ranks" in r.stderr and "torch.distributed.run" in r.stderr, r.stderr[-2000:]
    if not torch.cuda.is_available():
        assert "needs an MI355X" in r.stderr, r.stderr[-2000:]              # said by the rank, not by the parent
        assert r.returncode != 0                                            # ... and its failure is the parent's exit code
        assert r.stdout.strip() == ""                                       # no JSON line was invented


def _clock_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    steps = 0
    # rank 1's clock runs out two steps before rank 0's: both must stop after the same number of (collective) steps
    while not bench.any_rank_says(steps >= (5 if rank == 0 else 3), torch.device("cpu")):
        t = torch.ones(4)
        dist.all_reduce(t)                           # the "step": hangs if the partner has left the loop
        steps += 1
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(out, torch.tensor([steps]))
    if rank == 0:
        q.put([int(x) for x in out])
    dist.barrier()
    dist.destroy_process_group()


def test_bench_preheat_loop_ends_on_every_rank_after_the_same_number_of_steps():
    """bench.py's pre-heat runs "for N seconds" by each rank's own clock; its steps contain collectives (DDP all-reduce,
    num_boxes), so the ranks have to agree on the moment to stop (bench.any_rank_says) -- otherwise the N > 1 line never
    appears."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_clock_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() == [3, 3]


def _control_path_worker(rank, world, port, q):
    """bench.py's N-rank control path with a stand-in step (one all-reduce + a rank-dependent sleep): the pre-heat agreement, the
    barriers around the timed region, the max-over-ranks reduction and the whole-job value -- the very functions main() calls."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    cpu = torch.device("cpu")
    calls = {"n": 0}

    def step(batch):
        t = torch.ones(8) * batch
        dist.all_reduce(t)                            # a step is a set of collectives: hangs if a partner has left the loop
        assert t[0].item() == batch * world
        time.sleep(0.002 * (rank + 1))                # rank 7 is the slowest
        calls["n"] += 1

    # every rank's own clock would stop it at a different step; all must stop together
    n, _ = bench.preheat(step, 1.0, 0.05 * (rank + 1), cpu, lambda: None)
    assert calls["n"] == n
    steps = 6
    elapsed = bench.timed_steps(step, 1.0, steps, cpu)
    assert calls["n"] == n + steps                    # EXACTLY K timed steps
    out = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, torch.tensor([float(n), elapsed], dtype=torch.float64))
    if rank == 0:
        q.put([[float(x[0]), float(x[1])] for x in out])
    dist.barrier()
    dist.destroy_process_group()


def test_bench_control_path_with_eight_ranks():
    """N = 8 over gloo on the CPU: what the driver's 8-GPU run exercises besides RCCL itself.  Every rank ends the pre-heat after the
    same number of steps, times exactly K steps between barriers and reports the SAME (slowest rank's) time."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_control_path_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    rows = q.get()
    assert len({r[0] for r in rows}) == 1 and rows[0][0] >= 1, rows          # same pre-heat step count everywhere
    assert len({r[1] for r in rows}) == 1, rows                              # one time: the maximum over ranks
    assert rows[0][1] >= 6 * 0.002 * world * 0.9, rows                       # ... which is at least the slowest rank's sleeps


def test_bench_rehearsal_refuses_more_than_two_ranks():
    """MONOSOWA_BENCH_REHEARSAL=1 (every rank on device 0, gloo) is a two-rank tool: with more ranks it only measures gloo's
    host-staged all-reduce (round 3: 23.6 s per step at three ranks, no step in 400 s at four).  bench.py says so instead of burning
    GPU minutes -- checked on the text, the guard sits behind the "needs an MI355X" test."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    guard = src.index("supports WORLD_SIZE <= 2")
    assert src.rindex("if world > 2:", 0, guard) > src.index('rehearsal = os.environ.get("MONOSOWA_BENCH_REHEARSAL")')
