"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REFERENCE's own Python.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [msda] [module] [transformer] [depth] [misc]

The reference tree is imported read-only, unmodified.  Third-party symbols that are absent
from this image are shimmed (never reference code): the unbuilt CUDA extension module
``MultiScaleDeformableAttention``, ``torchvision`` (only ``__version__`` and
``ops.boxes.box_area``), and the torch<1.9 name ``_LinearWithBias`` that the reference's
version test selects under torch 2.x (ops/modules/ms_deform_attn.py:34).  The fixtures are
data only: inputs, expected outputs, seeded state dicts.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/MonoDETR"
MD = REF + "/lib/models/monodetr"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ----------------------------------------------------------------------------- reference import
def _shims():
    sys.dont_write_bytecode = True
    if "MultiScaleDeformableAttention" not in sys.modules:
        sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.__version__ = "0.14.1"
        ops = types.ModuleType("torchvision.ops")
        boxes = types.ModuleType("torchvision.ops.boxes")
        boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        ops.boxes = boxes
        tv.ops = ops
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": boxes})
    import torch.nn.modules.linear as lin
    if not hasattr(lin, "_LinearWithBias"):
        lin._LinearWithBias = lin.NonDynamicallyQuantizableLinear
    if "torch._overrides" not in sys.modules:
        sys.modules["torch._overrides"] = torch.overrides


def ref_core():
    """The reference's ms_deform_attn_core_pytorch, imported the way ops/test.py:18 does."""
    _shims()
    if MD + "/ops" not in sys.path:
        sys.path.insert(0, MD + "/ops")
    from functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    return ms_deform_attn_core_pytorch


def ref_pkg():
    """Synthetic parent package so reference sub-modules import without lib/models/monodetr/__init__."""
    _shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "mdpkg" not in sys.modules:
        pkg = types.ModuleType("mdpkg")
        pkg.__path__ = [MD]
        sys.modules["mdpkg"] = pkg
    import importlib
    msda_mod = importlib.import_module("mdpkg.ops.modules.ms_deform_attn")
    core = importlib.import_module("mdpkg.ops.functions.ms_deform_attn_func").ms_deform_attn_core_pytorch

    class _Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return core(value, shapes, loc, w)
    msda_mod.MSDeformAttnFunction = _Fn
    return importlib


# ----------------------------------------------------------------------------- helpers
def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **_np(arrays))
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def _msda_case(core, name, seed, B, M, D, Lq, shapes, P, dtype, loc_range=(0.0, 1.0), value_scale=0.01):
    torch.manual_seed(seed)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = (torch.rand(B, S, M, D) * value_scale).to(dtype)
    lo, hi = loc_range
    loc = (torch.rand(B, Lq, M, L, P, 2) * (hi - lo) + lo).to(dtype)
    w = torch.rand(B, Lq, M, L, P) + 1e-5
    w = (w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(B, Lq, M * D).to(dtype)
    value.requires_grad_(True)
    loc.requires_grad_(True)
    w.requires_grad_(True)
    out = core(value, shapes, loc, w)
    out.backward(grad_out)
    _save(name, value=value, shapes=shapes, lsi=_lsi(shapes), loc=loc, attw=w, grad_out=grad_out,
          out=out, grad_value=value.grad, grad_loc=loc.grad, grad_attw=w.grad)


def gen_msda():
    core = ref_core()
    # (i) the geometry of the reference's own ops/test.py:21-36 (N=1,M=2,D=2,Lq=2,L=2,P=2, seed 3)
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        _msda_case(core, "msda_optest_" + tag, 3, 1, 2, 2, 2, [(6, 4), (3, 2)], 2, dt)
    # (i') the D list of ops/test.py:85 that exercises every reference backward variant (small ones)
    for D in (30, 32, 64, 71):
        _msda_case(core, "msda_optest_D%d_f64" % D, 3, 1, 2, D, 2, [(6, 4), (3, 2)], 2, torch.float64)
    # (ii) shipped head geometry M=8,D=32,L=4,P=4 on reduced levels, out-of-range locations
    lv = [(6, 20), (3, 10), (2, 5), (1, 3)]
    _msda_case(core, "msda_d32_q50_f32", 11, 2, 8, 32, 50, lv, 4, torch.float32, (-0.25, 1.25), 1.0)
    _msda_case(core, "msda_d32_q110_f32", 12, 1, 8, 32, 110, lv, 4, torch.float32, (-0.25, 1.25), 1.0)
    _msda_case(core, "msda_d32_q50_f64", 13, 1, 8, 32, 50, lv, 4, torch.float64, (-0.25, 1.25), 1.0)
    # (ii') ragged / edge: a 1x1 level, a single query, locations exactly on the borders
    _msda_case(core, "msda_edge_f64", 14, 2, 3, 5, 1, [(1, 1), (2, 7), (5, 1)], 3, torch.float64, (-0.1, 1.1), 1.0)


if __name__ == "__main__":
    which = sys.argv[1:] or ["msda"]
    for w in which:
        globals()["gen_" + w]()
