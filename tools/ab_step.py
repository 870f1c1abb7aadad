"""Paired A/B timing of one module-level setting inside ONE process: the train step alternates between the two values
(A, B, A, B, ...), every step is timed with a device sync, medians are compared.  Box-to-box and run-to-run noise
(+-2 %) cancels; differences of 0.3 % become visible.

    python tools/ab_step.py monosowa_amd.pointwise.LN_MIN_ROWS 1 16384 [--steps 60]
    python tools/ab_step.py option:scatter_sorted 0 2          (msda_set_option switches)
"""
import importlib
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd import miopen_tuning   # noqa: E402
miopen_tuning.use_shipped_db(0)

import torch   # noqa: E402
import yaml    # noqa: E402

from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout   # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer  # noqa: E402
from monosowa_amd.monodetr.criterion import weighted_total   # noqa: E402
from monosowa_amd.synthetic import make_batch, prepare_targets    # noqa: E402


def main():
    path, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 60
    if path.startswith("env:"):               # an environment variable read per call by the native code: env:MSDA_GATHER_CHUNKS 8 16
        class _Env:
            pass
        mod, attr, values = _Env(), path.split(":", 1)[1], [a, b]
        _Env.__setattr__ = lambda self, k, v: os.environ.__setitem__(k, str(v))
    elif path.startswith("option:"):            # a kernel-generation switch of the MSDA library: option:scatter_sorted 0 2
        from monosowa_amd import _lib

        class _Opt:
            pass
        mod, attr, values = _Opt(), path.split(":", 1)[1], [int(a), int(b)]
        _Opt.__setattr__ = lambda self, k, v: _lib.set_option(k, v)
    else:
        mod_name, attr = path.rsplit(".", 1)
        mod = importlib.import_module(mod_name)
        cast = type(getattr(mod, attr))
        values = [cast(a) if cast is not bool else a == "1", cast(b) if cast is not bool else b == "1"]
    dev = torch.device("cuda:0")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "monodetr.yaml")))
    model, crit = build_model(cfg["model"])
    model = to_mi355x_layout(model.to(dev)).train()
    crit.to(dev).train()
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, info = make_batch(16, dev)
    inputs = inputs.contiguous(memory_format=torch.channels_last)

    def step():
        tl = prepare_targets(targets, 16)
        opt.zero_grad(set_to_none=True)
        o = model(inputs, calibs, tl, targets["img_size"])
        weighted_total(crit(o, tl), crit.weight_dict).backward()
        opt.step()
    for v in values * 8:                 # warm both variants
        setattr(mod, attr, v)
        step()
    torch.cuda.synchronize()
    block = int(sys.argv[sys.argv.index("--block") + 1]) if "--block" in sys.argv else 1
    times = {0: [], 1: []}
    for i in range(steps // block):
        k = i & 1
        setattr(mod, attr, values[k])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(block):           # block > 1: no sync between steps, as in bench.py (cross-step overlap counts)
            step()
        torch.cuda.synchronize()
        times[k].append((time.perf_counter() - t) * 1e3 / block)
    for k in (0, 1):
        ts = sorted(times[k])
        print("%s = %-8r median %.3f ms  mean %.3f  min %.3f  (n=%d)" % (attr, values[k], statistics.median(ts), statistics.mean(ts), ts[0], len(ts)), flush=True)
    print("B - A (median): %+.3f ms" % (statistics.median(times[1]) - statistics.median(times[0])), flush=True)


if __name__ == "__main__":
    main()
