"""host time of the criterion's tail (from the matcher's indices to the weighted total) and of backward()'s enqueue"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd import miopen_tuning
miopen_tuning.use_shipped_db(0)
import torch, yaml
from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
from monosowa_amd.helpers.optimizer_helper import build_optimizer
from monosowa_amd.monodetr import criterion as C
from monosowa_amd.synthetic import make_batch, prepare_targets

dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "..", "configs", "monodetr.yaml")))
model, crit = build_model(cfg["model"])
model = to_mi355x_layout(model.to(dev)).train(); crit.to(dev).train()
opt = build_optimizer(cfg["optimizer"], model)
inputs, calibs, targets, info = make_batch(16, dev)
inputs = inputs.contiguous(memory_format=torch.channels_last)
marks = {}
orig_end = crit.matcher.match_layers_end_flat
def end_flat(p):
    r = orig_end(p); marks["sync"] = time.perf_counter(); return r
crit.matcher.match_layers_end_flat = end_flat
acc = {"crit_tail": [], "bwd": [], "step": []}
for it in range(25):
    tl = prepare_targets(targets, 16)
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    o = model(inputs, calibs, tl, targets["img_size"])
    tot = None          # (drop the previous step's graph skeleton HERE, not in the assignment below: ~0.4 ms of teardown)
    tot = C.weighted_total(crit(o, tl), crit.weight_dict)
    t1 = time.perf_counter()
    tot.backward()
    t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if it >= 5:
        acc["crit_tail"].append((t1 - marks["sync"]) * 1e3); acc["bwd"].append((t2 - t1) * 1e3); acc["step"].append((t3 - t0) * 1e3)
import statistics
print({k: round(statistics.median(v), 3) for k, v in acc.items()})
