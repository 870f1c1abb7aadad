"""How long does the host take to enqueue one train step, against the GPU time of the step?"""
import os
import sys
import time

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
    marks = {}

    def step():
        t = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        o = model(inputs, calibs, tl, targets["img_size"])
        t1 = time.perf_counter()
        ld = crit(o, tl)
        tot = weighted_total(ld, crit.weight_dict)
        t2 = time.perf_counter()
        tot.backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        for k, v in (("fwd", t1 - t), ("crit", t2 - t1), ("bwd", t3 - t2), ("opt", t4 - t3)):
            marks[k] = marks.get(k, 0.0) + v
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if "--cached-match" in sys.argv:          # upper bound of what hiding the matcher's host round trip could give
        from monosowa_amd.monodetr.matcher import HungarianMatcher
        real = HungarianMatcher.match_layers
        cache = {}

        def cached(self, *a, **k):
            if "m" not in cache:
                cache["m"] = real(self, *a, **k)
            return cache["m"]
        HungarianMatcher.match_layers = cached
        step()
        torch.cuda.synchronize()
    marks.clear()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("host enqueue %.1f ms/step, wall %.1f ms/step; host split: %s" % (
        t_cpu / n * 1e3, t_all / n * 1e3, {k: round(v / n * 1e3, 1) for k, v in marks.items()}), flush=True)


if __name__ == "__main__":
    main()
