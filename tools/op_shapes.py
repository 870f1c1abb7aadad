"""Which ATen ops (by input shape) own the glue-kernel time of a train step?  torch.profiler, one step, stdout."""
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

    def step():
        opt.zero_grad(set_to_none=True)
        o = model(inputs, calibs, tl, targets["img_size"])
        ld = crit(o, tl)
        tot = weighted_total(ld, crit.weight_dict)
        tot.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print("warm", flush=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    want = sys.argv[1:] or ["aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::index_put_", "aten::_index_put_impl_",
                            "aten::fill_", "aten::zero_", "aten::cat", "aten::clamp", "aten::upsample_bilinear2d"]
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in want or "--all" in sys.argv]
    want = [w for w in want if w != "--all"]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:110 if '--all' in sys.argv else 70]:
        print("%-26s %8.1f us %4d x  %s" % (e.key, e.self_device_time_total, e.count, str(e.input_shapes)[:150]), flush=True)


if __name__ == "__main__":
    main()
