"""the kernels the criterion's forward launches in one train step, in order (torch.profiler)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yaml
from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
from monosowa_amd.monodetr.criterion import weighted_total
from monosowa_amd.synthetic import make_batch, prepare_targets
from torch.profiler import profile, ProfilerActivity, record_function

dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "..", "configs", "monodetr.yaml")))
model, crit = build_model(cfg["model"])
model = to_mi355x_layout(model.to(dev)).train(); crit.to(dev).train()
inputs, calibs, targets, info = make_batch(16, dev)
inputs = inputs.contiguous(memory_format=torch.channels_last)
tl = prepare_targets(targets, 16)
which = sys.argv[1] if len(sys.argv) > 1 else "criterion"
def step(prof=False):
    with record_function("STAGE_forward"):
        o = model(inputs, calibs, tl, targets["img_size"])
    with record_function("STAGE_criterion"):
        tot = weighted_total(crit(o, tl), crit.weight_dict)
    tot.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type != torch.autograd.DeviceType.CUDA]
stage = [e for e in ev if e.name == "STAGE_" + which][0]
inside = [e for e in ev if e.kernels and e.thread == stage.thread and e.time_range.start >= stage.time_range.start and e.time_range.end <= stage.time_range.end
          and not any(c.kernels for c in e.cpu_children) and not e.name.startswith("STAGE_")]
inside.sort(key=lambda e: e.time_range.start)
mods = [e for e in ev if e.name.startswith("MOD:")]
for e in inside:
    print("%8.1f us  %-28s %-60s %s" % ((e.time_range.start - stage.time_range.start), e.name[:28], e.kernels[0].name[:60], str(e.input_shapes)[:90]))
print(len(inside), "launching ops")
