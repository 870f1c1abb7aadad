"""Deterministic parameter fill shared by oracle/gen_golden.py (applied to the REFERENCE's modules)
and the tests (applied to this repo's modules): tensors are generated from the sorted state-dict key
order, so two modules with identical key names/shapes get identical values and a fixture only has to
store inputs, outputs and the key list -- which also pins state-dict compatibility."""
import json

import torch


def fill_deterministic(module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            if not t.dtype.is_floating_point:
                continue
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            leaf = name.rsplit(".", 1)[-1]
            if "running_var" in leaf:
                v = r.abs() + 0.5
            elif name.endswith("sampling_offsets.bias"):
                v = r * 3.0                               # a few pixels of offset
            elif name.endswith("depth_bin_values"):
                continue                                   # derived constant, keep
            elif t.dim() <= 1:
                is_norm_scale = leaf == "weight"
                v = 1.0 + 0.1 * r if is_norm_scale else 0.1 * r
            else:
                fan_in = t[0].numel()
                v = r * (1.0 / max(fan_in, 1) ** 0.5)
            t.copy_(v.to(t.dtype))
    return module


def key_manifest(module):
    return json.dumps({k: list(v.shape) for k, v in module.state_dict().items()}, sort_keys=True)
