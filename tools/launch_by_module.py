"""Kernel launches of one train step by where they come from: forward launches per module (depth <= --depth), backward launches
per autograd node type, criterion / optimizer as wholes.  stdout.   python tools/launch_by_module.py [--depth 3]"""
import collections
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout   # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer  # noqa: E402
from monosowa_amd.monodetr.criterion import weighted_total   # noqa: E402
from monosowa_amd.synthetic import make_batch, prepare_targets    # noqa: E402
from torch.profiler import profile, ProfilerActivity, record_function  # noqa: E402


def main():
    depth = int(sys.argv[sys.argv.index("--depth") + 1]) if "--depth" in sys.argv else 3
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

    ranges = {}
    for name, mod in model.named_modules():
        if name and name.count(".") < depth:
            def pre(m, a, _n=name):
                r = record_function("MOD:" + _n)
                r.__enter__()
                ranges.setdefault(id(m), []).append(r)
            def post(m, a, o):
                ranges[id(m)].pop().__exit__(None, None, None)
            mod.register_forward_pre_hook(pre)
            mod.register_forward_hook(post)

    def step():
        opt.zero_grad(set_to_none=True)
        with record_function("STAGE_forward"):
            o = model(inputs, calibs, tl, targets["img_size"])
        with record_function("STAGE_criterion"):
            tot = weighted_total(crit(o, tl), crit.weight_dict)
        with record_function("STAGE_backward"):
            tot.backward()
        with record_function("STAGE_optimizer"):
            opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type != torch.autograd.DeviceType.CUDA]
    scopes = [e for e in ev if e.name.startswith("MOD:") or e.name.startswith("STAGE_") or e.name.startswith("autograd::engine::evaluate_function")]
    scopes.sort(key=lambda e: (e.time_range.start, -e.time_range.end))
    launchers = [e for e in ev if e.kernels and not e.name.startswith(("MOD:", "STAGE_", "autograd::engine"))]
    # innermost scope by containment (same thread or not: the backward runs on its own thread)
    count, time = collections.Counter(), collections.Counter()
    import bisect
    starts = [s.time_range.start for s in scopes]
    for e in launchers:
        if any(c.kernels for c in e.cpu_children):
            continue                                          # count at the leaf that owns the launch
        i = bisect.bisect_right(starts, e.time_range.start) - 1
        best = None
        while i >= 0:
            s = scopes[i]
            if s.time_range.end >= e.time_range.end and s.thread == e.thread:
                best = s
                break
            i -= 1
        key = best.name if best is not None else "(none)"
        key = key.replace("autograd::engine::evaluate_function: ", "BWD ")
        count[key] += len(e.kernels)
        time[key] += sum(k.duration for k in e.kernels)
    print("launches by innermost scope (forward: module; backward: autograd node)")
    for k, n in count.most_common(90):
        print("%5d launches %9.1f us   %s" % (n, time[k], k))
    print("total", sum(count.values()))


if __name__ == "__main__":
    main()
