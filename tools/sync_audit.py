"""Lists every implicit device->host synchronisation of one train step (torch.cuda.set_sync_debug_mode("warn"))."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402
import yaml    # noqa: E402

from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout   # noqa: E402
from monosowa_amd.helpers.optimizer_helper import build_optimizer  # noqa: E402
from monosowa_amd.monodetr.criterion import weighted_total   # noqa: E402
from monosowa_amd.synthetic import make_batch, prepare_targets    # noqa: E402

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


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
print("%d synchronising calls in one step" % len(w))
for x in w:
    print("  %s:%d  %s" % (os.path.relpath(x.filename), x.lineno, str(x.message)[:100]))
