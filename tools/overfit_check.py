"""End-to-end sanity of the optimised train step: repeat one synthetic batch and watch the weighted loss fall."""
import os
import sys

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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    dev = torch.device("cuda:0")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "monodetr.yaml")))
    torch.manual_seed(444)
    model, crit = build_model(cfg["model"])
    model = to_mi355x_layout(model.to(dev)).train()
    crit.to(dev).train()
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, info = make_batch(16, dev)
    inputs = inputs.contiguous(memory_format=torch.channels_last)
    hist = []
    for i in range(steps):
        tl = prepare_targets(targets, 16)
        opt.zero_grad(set_to_none=True)
        losses = crit(model(inputs, calibs, tl, targets["img_size"]), tl)
        total = weighted_total(losses, crit.weight_dict)
        total.backward()
        opt.step()
        if i % 10 == 0 or i == steps - 1:
            hist.append(total.item())
            print("step %4d  weighted loss %.4f  (ce %.3f, bbox %.3f, depth %.3f, depth_map %.3f)" % (
                i, hist[-1], losses["loss_ce"].item(), losses["loss_bbox"].item(), losses["loss_depth"].item(),
                losses["loss_depth_map"].item()), flush=True)
    assert all(h == h for h in hist), "NaN in the loss"
    assert hist[-1] < 0.7 * hist[0], "the loss did not fall: %s" % hist
    print("OK: %.3f -> %.3f" % (hist[0], hist[-1]))


if __name__ == "__main__":
    main()
