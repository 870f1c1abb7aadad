"""host time of the pieces of the criterion's tail (behind the matcher's indices), measured by wrapping them"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from monosowa_amd import miopen_tuning
miopen_tuning.use_shipped_db(0)
import torch, yaml
from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
from monosowa_amd.helpers.optimizer_helper import build_optimizer
from monosowa_amd.monodetr import criterion as C
from monosowa_amd import pointwise as PW
from monosowa_amd.synthetic import make_batch, prepare_targets
dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "..", "configs", "monodetr.yaml")))
model, crit = build_model(cfg["model"])
model = to_mi355x_layout(model.to(dev)).train(); crit.to(dev).train()
opt = build_optimizer(cfg["optimizer"], model)
inputs, calibs, targets, info = make_batch(16, dev)
inputs = inputs.contiguous(memory_format=torch.channels_last)
acc = collections.defaultdict(list)
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label].append((time.perf_counter() - t0) * 1e6); return r
    setattr(obj, name, g)
wrap(PW._FocalClassification, "apply", "focal apply")
wrap(PW._MatchedLosses, "apply", "matched apply")
C.SetCriterion._finish = staticmethod((lambda f: (lambda *a, **k: (lambda t0, r: (acc["_finish"].append((time.perf_counter() - t0) * 1e6), r)[1])(time.perf_counter(), f(*a, **k))))(C.SetCriterion._finish))
wrap(C, "weighted_total", "weighted_total")
import monosowa_amd.monodetr.criterion as CC
_orig_sup = CC.focal_classification_supported
def _sup(*a, **k):
    acc["sync -> focal_supported (idx copy)"].append((time.perf_counter() - marks["sync"]) * 1e6); return _orig_sup(*a, **k)
CC.focal_classification_supported = _sup
orig_end = crit.matcher.match_layers_end_flat
marks = {}
def end_flat(p):
    r = orig_end(p); marks["sync"] = time.perf_counter(); return r
crit.matcher.match_layers_end_flat = end_flat
import gc
if os.environ.get("NOGC"): gc.disable()
for it in range(25):
    tl = prepare_targets(targets, 16)
    opt.zero_grad(set_to_none=True)
    o = model(inputs, calibs, tl, targets["img_size"])
    ld = crit(o, tl)
    t1 = time.perf_counter()
    tot = C.weighted_total(ld, crit.weight_dict)
    t2 = time.perf_counter()
    tot.backward()
    opt.step()
    if it >= 5:
        acc["after sync -> criterion returns"].append((t1 - marks["sync"]) * 1e6)
        acc["after sync -> total"].append((t2 - marks["sync"]) * 1e6)
        acc["t2 - t1"].append((t2 - t1) * 1e6)
        if it >= 22: print("it", it, "t1-sync %.0f  t2-t1 %.0f  inner wt %.0f" % ((t1 - marks["sync"]) * 1e6, (t2 - t1) * 1e6, acc["weighted_total"][-1]))
torch.cuda.synchronize()
import statistics
for k, v in acc.items():
    print("%-34s %7.1f us" % (k, statistics.median(v[-15:])))
