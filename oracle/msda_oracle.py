"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle for MSDA.

Two independent restatements of the reference operator:

* :func:`forward` / :func:`backward` -- ctypes binding of ``msda_oracle.c`` (scalar C
  restatement of the reference device code, ``ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,
  237-403``), f32 and f64.
* :func:`msda_core_torch` -- per-level ``grid_sample`` formulation, the algorithm of the
  reference's only CPU-capable definition ``ms_deform_attn_core_pytorch``
  (``ops/functions/ms_deform_attn_func.py:41-61``).  Used as the CPU baseline ("port").

Both are pinned against golden vectors captured from the reference's own Python
(``oracle/gen_golden.py`` -> ``tests/golden/msda_*.npz``) by ``tests/test_oracle_golden.py``.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile ``libmsda_oracle.so`` with gcc (seconds)."""
    so = os.path.join(_HERE, "libmsda_oracle.so")
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmsda_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attw):
    dt = np.float64 if np.asarray(value).dtype == np.float64 else np.float32
    value = np.ascontiguousarray(value, dtype=dt)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attw = np.ascontiguousarray(attw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    assert loc.shape == (B, Lq, M, L, P, 2) and attw.shape == (B, Lq, M, L, P)
    assert shapes.shape == (L, 2) and lsi.shape == (L,)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return dt, value, shapes, lsi, loc, attw, (B, S, M, D, L, Lq, P)


def forward(value, shapes, lsi, loc, attw):
    """numpy in, numpy out ``[B, Lq, M*D]`` (same dtype as ``value``)."""
    dt, value, shapes, lsi, loc, attw, dims = _prep(value, shapes, lsi, loc, attw)
    B, S, M, D, L, Lq, P = dims
    out = np.zeros((B, Lq, M * D), dtype=dt)
    fn = getattr(_lib(), "msda_oracle_forward_" + ("f64" if dt == np.float64 else "f32"))
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attw),
       *[ctypes.c_int(x) for x in dims], _ptr(out))
    return out


def backward(value, shapes, lsi, loc, attw, grad_out):
    """Returns ``(grad_value, grad_loc, grad_attw)`` shaped like the inputs."""
    dt, value, shapes, lsi, loc, attw, dims = _prep(value, shapes, lsi, loc, attw)
    grad_out = np.ascontiguousarray(grad_out, dtype=dt)
    gv, gl, gw = np.zeros_like(value), np.zeros_like(loc), np.zeros_like(attw)
    fn = getattr(_lib(), "msda_oracle_backward_" + ("f64" if dt == np.float64 else "f32"))
    fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attw), _ptr(grad_out),
       *[ctypes.c_int(x) for x in dims], _ptr(gv), _ptr(gl), _ptr(gw))
    return gv, gl, gw


def msda_core_torch(value, shapes, loc, attw):
    """Per-level ``grid_sample`` formulation (differentiable, any device torch supports).

    Follows ``ms_deform_attn_core_pytorch`` (ms_deform_attn_func.py:41-61): split value by
    level, map locations to [-1, 1], bilinear/zeros/align_corners=False sampling, weight, sum.
    """
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in shapes.tolist()]
    pieces = value.split([h * w for h, w in hw], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        v = pieces[lvl].flatten(2).transpose(1, 2).reshape(B * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = attw.transpose(1, 2).reshape(B * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def level_start_index(shapes):
    shapes = np.asarray(shapes, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int64)
