"""Who launches the small kernels of a train step?  One profiled step (torch.profiler): ATen / custom ops that own device
time, sorted by launch count, then the same grouped by input shape for the busiest ops.  stdout."""
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout   # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer  # noqa: E402
from monosowa_amd.monodetr.criterion import weighted_total   # noqa: E402
from monosowa_amd.synthetic import make_batch, prepare_targets    # noqa: E402


def main():
    B = 16
    dev = torch.device("cuda:0")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "monodetr.yaml")))
    model, crit = build_model(cfg["model"])
    model.to(dev).train()
    to_mi355x_layout(model)
    crit.to(dev).train()
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, info = make_batch(B, dev)
    inputs = inputs.contiguous(memory_format=torch.channels_last)
    tl = prepare_targets(targets, B)
    from torch.profiler import profile, ProfilerActivity, record_function

    def step():
        opt.zero_grad(set_to_none=True)
        with record_function("STAGE_forward"):
            o = model(inputs, calibs, tl, targets["img_size"])
        with record_function("STAGE_criterion"):
            ld = crit(o, tl)
            tot = weighted_total(ld, crit.weight_dict)
        with record_function("STAGE_backward"):
            tot.backward()
        with record_function("STAGE_optimizer"):
            opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack="--stack" in sys.argv) as prof:
        step()
        torch.cuda.synchronize()
    ev = prof.events()
    kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
    print("device kernels in the step: %d, %.2f ms" % (len(kernels), sum(k.device_time for k in kernels) / 1e3))
    stages = [e for e in ev if e.name.startswith("STAGE_")]
    for s in stages:
        ks = 0
        for e in ev:
            if e.device_type != torch.autograd.DeviceType.CUDA and e.time_range.start >= s.time_range.start and e.time_range.end <= s.time_range.end:
                ks += len(e.kernels) if not e.cpu_children else 0
        print("%-18s launches %5d   host %.1f ms" % (s.name, ks, (s.time_range.end - s.time_range.start) / 1e3))
    rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith("STAGE_")]
    rows.sort(key=lambda e: -e.count)
    print("\n-- ops owning device time, by launch count")
    for e in rows[:60]:
        print("%-60s %5d x %9.1f us" % (e.key[:60], e.count, e.self_device_time_total))
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and not e.key.startswith("STAGE_")]
    rows.sort(key=lambda e: -e.count)
    print("\n-- the same by input shape")
    for e in rows[:120]:
        print("%-34s %4d x %8.1f us  %s" % (e.key[:34], e.count, e.self_device_time_total, str(e.input_shapes)[:140]))
    if "--stack" in sys.argv:
        stacks(prof)
    if "--by-time" in sys.argv:
        print("\n-- glue ops by input shape, by device time")
        names = ("aten::sum", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::cat", "aten::div", "aten::sub", "aten::clamp",
                 "aten::index", "aten::gather", "aten::where", "aten::sigmoid", "aten::exp", "aten::neg", "aten::mean", "aten::zero_")
        rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in names and e.self_device_time_total > 0]
        rows.sort(key=lambda e: -e.self_device_time_total)
        for e in rows[:70]:
            print("%-14s %4d x %8.1f us  %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:150]))




def stacks(prof, names=("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::cat", "aten::div", "aten::sub")):
    """--stack: the innermost frames of this repo behind the small glue ops, by launch count."""
    rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key in names and e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.count)
    print("\n-- glue ops by call site")
    for e in rows[:60]:
        mine = [f for f in e.stack if "monosowa_amd" in f or "bench" in f or "tools" in f][:2]
        print("%-14s %4d x %8.1f us  %s" % (e.key, e.count, e.self_device_time_total, " <- ".join(m.strip()[-90:] for m in mine)))


if __name__ == "__main__":
    main()
