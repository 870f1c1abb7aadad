"""Which ATen ops put the small kernels on the GPU timeline of one train step: torch.profiler over 3 steps of bench.py's default
workload, grouped by (op, input shapes, innermost monosowa_amd frame).  Run on the GPU box:
    python tools/debug/op_census.py [--ops cat,fill_,zero_,add,add_,copy_,stack,zeros,zeros_like] > gpurun_out/op_census.txt"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="cat,fill_,zero_,add,add_,copy_,stack,zeros,zeros_like,sum,mul,clone,contiguous,index_select,masked_fill_,masked_fill")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import bench
    sys.argv = ["bench.py"]
    args = bench.parse()
    from monosowa_amd import miopen_tuning
    miopen_tuning.use_shipped_db(0)
    import torch
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    from monosowa_amd.synthetic import make_batch
    cfg, model, criterion, optimizer, (W, H) = bench.build_everything(args, device)
    model.train(True)
    criterion.train(True)
    batch = make_batch(args.batch, device, seed=444, resolution=(W, H), mixed_cameras=False)
    batch = (batch[0].contiguous(memory_format=torch.channels_last),) + batch[1:]
    step = bench.train_step_fn(model, criterion, optimizer)
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(a.steps):
            step(batch)
        torch.cuda.synchronize()
    want = set("aten::" + o for o in a.ops.split(","))
    groups = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name not in want:
            continue
        dev_us = getattr(ev, "device_time_total", None)
        if dev_us is None:
            dev_us = getattr(ev, "cuda_time_total", 0.0)
        own = sum(getattr(k, "device_time", getattr(k, "cuda_time", 0.0)) if hasattr(k, "device_time") or hasattr(k, "cuda_time") else 0.0 for k in ev.kernels) if ev.kernels else 0.0
        if not ev.kernels:
            continue                                   # only ops that launch something themselves
        frame = "?"
        for s in (ev.stack or []):
            if "monosowa_amd" in s or "bench.py" in s:
                frame = s.split("monosowa_amd/")[-1] if "monosowa_amd/" in s else s
                break
        if frame == "?" and ev.stack:
            frame = "(autograd) " + ev.stack[0][-60:] if ev.stack else "?"
        shapes = str(ev.input_shapes)[:90]
        g = groups[(ev.name, shapes, frame)]
        g[0] += len(ev.kernels)
        g[1] += own
    rows = sorted(groups.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows) / a.steps
    print("launches/step from the listed ops: %.1f" % tot)
    for (name, shapes, frame), (n, us) in rows[:120]:
        print("%6.1f /step %8.1f us/step  %-18s %-92s %s" % (n / a.steps, us / a.steps, name, shapes, frame))


if __name__ == "__main__":
    main()
