"""Where does a MonoDETR train step spend its time?  Prints as it goes (stdout, flushed)."""
import os
import sys
import time

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monosowa_amd.helpers.model_helper import build_model        # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer  # noqa: E402
from monosowa_amd.monodetr.criterion import weighted_total   # noqa: E402
from monosowa_amd.synthetic import make_batch, prepare_targets    # noqa: E402

T0 = time.time()


def say(*a):
    print("[%7.1fs]" % (time.time() - T0), *a, flush=True)


def timed(name, fn, n=3):
    torch.cuda.synchronize()
    t = time.time()
    out = fn()
    torch.cuda.synchronize()
    first = time.time() - t
    t = time.time()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    say("%-34s first %.3f s, then %.2f ms" % (name, first, (time.time() - t) / n * 1e3))
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.backends.cudnn.benchmark = "--find" in sys.argv
    dev = torch.device("cuda:0")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "monodetr.yaml")))
    model, crit = build_model(cfg["model"])
    model.to(dev).train()
    if "--cl" in sys.argv:
        model.backbone.to(memory_format=torch.channels_last)
    crit.to(dev).train()
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, info = make_batch(B, dev)
    if "--cl" in sys.argv:
        inputs = inputs.contiguous(memory_format=torch.channels_last)
    tl = prepare_targets(targets, B)
    say("built; B =", B)
    with torch.no_grad():
        timed("backbone fwd (no grad)", lambda: model.backbone(inputs))
    feats, pos = timed("backbone fwd (grad)", lambda: model.backbone(inputs))

    def bb_fb():
        f, _ = model.backbone(inputs)
        sum(x.tensors.sum() for x in f).backward()
    timed("backbone fwd+bwd", bb_fb)

    def full_fwd():
        return model(inputs, calibs, tl, targets["img_size"])
    out = timed("model fwd", full_fwd)
    timed("criterion", lambda: crit(out, tl))

    def step():
        opt.zero_grad(set_to_none=True)
        o = model(inputs, calibs, tl, targets["img_size"])
        ld = crit(o, tl)
        tot = weighted_total(ld, crit.weight_dict)
        tot.backward()
        opt.step()
    timed("full train step", step, n=5)
    say("max mem GB", torch.cuda.max_memory_allocated() / 2**30)


if __name__ == "__main__":
    main()
