"""Mirror of the reference's compiled extension module ``MultiScaleDeformableAttention``
(ops/src/vision.cpp:13-16) on top of the C-ABI HIP library.

Same two functions, same argument order and the same preconditions as the reference host code
(ops/src/cuda/ms_deform_attn_cuda.cu:28-52,93-119; ops/src/ms_deform_attn.h:38,60):

* every tensor contiguous and on the GPU, else ``RuntimeError``;
* CPU tensors -> ``RuntimeError("Not implemented on the CPU")``;
* ``batch % min(batch, im2col_step) == 0``;
* float32 / float64 values, int64 ``spatial_shapes`` / ``level_start_index``.

Differences: kernel launch errors raise (the reference printf's them, cuh:948-952), and the
whole batch goes out in one launch on ``torch.cuda.current_stream()`` (64-bit indexing makes the
reference's im2col_step slicing unnecessary; results are identical for every step).

To make reference code that does ``import MultiScaleDeformableAttention as MSDA``
(ops/functions/ms_deform_attn_func.py:18) pick this up, call :func:`install`.
"""
import sys

import os

import torch

from ._lib import on_device

from . import _lib

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward", "ms_deform_attn_fused_forward",
           "ms_deform_attn_fused_backward", "fused_supported", "install", "LaunchTimer"]


class LaunchTimer:
    """Optional per-launch timing of the C-ABI calls with HIP events recorded on the stream the kernels
    are launched on (bench.py's roofline leg).  Off by default: no events, no overhead.

        with LaunchTimer() as t: ...train steps...
        t.summary() -> {("fwd", Lq): {"launches": n, "avg_ms": x, "B": b, ...}, ...}
    """
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        LaunchTimer.active = self
        return self

    def __exit__(self, *exc):
        LaunchTimer.active = None

    def bracket(self, kind, dims):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        self.records.append((kind, dims, e0, e1))
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, dims, e0, e1 in self.records:
            d = out.setdefault((kind, dims), {"launches": 0, "total_ms": 0.0})
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
        return out


def _assert(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_common(named, im2col_step):
    value = named[0][1]
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")   # ms_deform_attn.h:38,60
    for name, t in named:
        _assert(t.is_contiguous(), "%s tensor has to be contiguous" % name)
        _assert(t.is_cuda, "%s must be a CUDA tensor" % name)
        _assert(t.device == value.device, "%s must be on %s" % (name, value.device))
    _assert(value.dtype in (torch.float32, torch.float64), "value must be float32 or float64")
    batch = value.size(0)
    step = min(batch, int(im2col_step))
    _assert(step > 0 and batch % step == 0, "batch(%d) must divide im2col_step(%d)" % (batch, step))


def _dims(value, spatial_shapes, sampling_loc, attn_weight):
    _assert(value.dim() == 4 and sampling_loc.dim() == 6 and attn_weight.dim() == 5, "bad tensor ranks")
    B, S, M, D = value.shape
    L = spatial_shapes.size(0)
    Lq, P = sampling_loc.size(1), sampling_loc.size(4)
    _assert(tuple(sampling_loc.shape) == (B, Lq, M, L, P, 2), "sampling_loc shape mismatch")
    _assert(tuple(attn_weight.shape) == (B, Lq, M, L, P), "attn_weight shape mismatch")
    _assert(spatial_shapes.dtype == torch.int64 and tuple(spatial_shapes.shape) == (L, 2), "spatial_shapes must be int64 [L,2]")
    return B, S, M, D, L, Lq, P


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step,
                           host_geom=None):
    """-> Tensor [B, Lq, M*D]   (reference: ms_deform_attn_cuda_forward, cu:20-80)"""
    _check_common([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)], im2col_step)
    B, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, attn_weight)
    _assert(level_start_index.dtype == torch.int64 and level_start_index.numel() == L, "level_start_index must be int64 [L]")
    _assert(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype, "dtype mismatch")
    lib = _lib.load()
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    if Lq == 0:        # the reference returns its (empty) at::zeros output: its launch of zero blocks fails and is only printed (cuh:948-952)
        return out
    fn = lib.msda_forward_f32 if value.dtype == torch.float32 else lib.msda_forward_f64
    if host_geom is None:      # only a pre-attached copy is used here: the forward never synchronises for it
        host_geom = getattr(spatial_shapes, "_msda_host_geometry", None) or (None, None)
    timer = LaunchTimer.active
    with on_device(value.device):
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("fwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        code = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                  sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
                  B, S, M, D, L, Lq, P, host_geom[0], host_geom[1], raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, "ms_deform_attn_forward")
    return out


def host_geometry(spatial_shapes, level_start_index):
    """Host copies of the (tiny) pyramid tensors for the backward's launch plan.  A caller that
    built the pyramid can pre-attach them (``attach_host_geometry``) and avoid the device->host sync."""
    cached = getattr(spatial_shapes, "_msda_host_geometry", None)
    if cached is not None:
        return cached
    return attach_host_geometry(spatial_shapes, level_start_index,
                                spatial_shapes.tolist(), level_start_index.tolist())


def attach_host_geometry(spatial_shapes, level_start_index, shapes_list, lsi_list):
    import ctypes
    L = len(shapes_list)
    sh = (ctypes.c_int64 * (2 * L))(*[int(x) for hw in shapes_list for x in hw])
    ls = (ctypes.c_int64 * L)(*[int(x) for x in lsi_list])
    geom = (sh, ls)
    try:
        spatial_shapes._msda_host_geometry = geom
    except Exception:
        pass
    return geom


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step,
                            host_geom=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]   (reference: cu:83-153)"""
    _check_common([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)], im2col_step)
    B, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, attn_weight)
    _assert(grad_output.numel() == B * Lq * M * D and grad_output.dtype == value.dtype, "grad_output shape/dtype mismatch")
    lib = _lib.load()
    if Lq == 0:        # no query: nothing is added to the reference's zero-initialised gradients (cu:121-123)
        return [torch.zeros_like(value), torch.zeros_like(sampling_loc), torch.zeros_like(attn_weight)]
    grad_value = torch.empty_like(value)
    grad_loc = torch.empty_like(sampling_loc)
    grad_w = torch.empty_like(attn_weight)
    ws_bytes = lib.msda_backward_workspace_bytes(B, S, M, D, L, Lq, P, value.element_size())
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device) if ws_bytes else None
    fn = lib.msda_backward_f32 if value.dtype == torch.float32 else lib.msda_backward_f64
    if host_geom is None:
        host_geom = host_geometry(spatial_shapes, level_start_index) if ws_bytes else (None, None)
    timer = LaunchTimer.active
    with on_device(value.device):
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("bwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        code = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                  sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                  grad_value.data_ptr(), grad_loc.data_ptr(), grad_w.data_ptr(),
                  B, S, M, D, L, Lq, P, host_geom[0], host_geom[1],
                  ws.data_ptr() if ws is not None else None, ws_bytes, raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_w]


def fused_supported(value, spatial_shapes, sampling_offsets, reference_points):
    """The fused operator covers the shipped geometry: float32 on the GPU, head dim 32, 4 levels x 4 points,
    reference points without gradient, and a pyramid whose host copy is attached."""
    return (value.is_cuda and value.dtype == torch.float32 and value.dim() == 4 and value.size(3) == 32
            and sampling_offsets.dim() == 6 and sampling_offsets.size(3) == 4 and sampling_offsets.size(4) == 4
            and value.size(2) * 32 <= 1024 and reference_points.size(-1) in (2, 6) and not reference_points.requires_grad
            and getattr(spatial_shapes, "_msda_host_geometry", None) is not None)


def _fused_dims(value, offsets, logits, ref):
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = offsets.shape
    _assert(tuple(logits.shape) in ((B, Lq, M, L, P), (B, Lq, M, L * P)), "logits shape mismatch")
    _assert(tuple(ref.shape[:3]) == (B, Lq, L) and ref.size(3) in (2, 6), "reference_points must be [B, Lq, L, 2|6]")
    for name, t in (("value", value), ("sampling_offsets", offsets), ("attention_logits", logits), ("reference_points", ref)):
        _assert(t.is_contiguous() and t.is_cuda and t.dtype == torch.float32, "%s must be a contiguous float32 CUDA tensor" % name)
    return B, S, M, D, L, Lq, P


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
                                 reference_points):
    """softmax + sampling-location arithmetic + sampling in one kernel (msda_fused_forward_f32) -> [B, Lq, M*D]."""
    B, S, M, D, L, Lq, P = _fused_dims(value, sampling_offsets, attention_logits, reference_points)
    geom = host_geometry(spatial_shapes, level_start_index)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    timer = LaunchTimer.active
    with on_device(value.device):
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("fwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        code = _lib.load().msda_fused_forward_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_offsets.data_ptr(),
            attention_logits.data_ptr(), reference_points.data_ptr(), reference_points.size(3), out.data_ptr(),
            B, S, M, D, L, Lq, P, geom[0], geom[1], raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, "ms_deform_attn_fused_forward")
    return out


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, sampling_offsets, attention_logits,
                                  reference_points, grad_output):
    """-> [grad_value, grad_sampling_offsets, grad_attention_logits]"""
    B, S, M, D, L, Lq, P = _fused_dims(value, sampling_offsets, attention_logits, reference_points)
    _assert(grad_output.is_contiguous() and grad_output.numel() == B * Lq * M * D, "grad_output shape mismatch")
    lib = _lib.load()
    geom = host_geometry(spatial_shapes, level_start_index)
    grad_value = torch.empty_like(value)
    grad_off = torch.empty_like(sampling_offsets)
    grad_logits = torch.empty_like(attention_logits)
    ws_bytes = lib.msda_backward_workspace_bytes(B, S, M, D, L, Lq, P, 4)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device)
    timer = LaunchTimer.active
    with on_device(value.device):
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("bwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        code = lib.msda_fused_backward_f32(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_offsets.data_ptr(),
            attention_logits.data_ptr(), reference_points.data_ptr(), reference_points.size(3), grad_output.data_ptr(),
            grad_value.data_ptr(), grad_off.data_ptr(), grad_logits.data_ptr(), B, S, M, D, L, Lq, P, geom[0], geom[1],
            ws.data_ptr(), ws_bytes, raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, "ms_deform_attn_fused_backward")
    return [grad_value, grad_off, grad_logits]


def install():
    """Register this module under the reference's import name."""
    sys.modules["MultiScaleDeformableAttention"] = sys.modules[__name__]
    return sys.modules[__name__]


def _value_view(value, value_mask):
    """(token stride in floats, mask pointer or None, mask tensor to keep alive) of a [B, S, M, D] value VIEW: dense, or a
    column block of a wider [B, S, C] projection (the last two dimensions dense, batch stride = S * token stride)."""
    B, S, M, D = value.shape
    ts = value.stride(1)
    _assert(value.dtype == torch.float32 and value.stride(3) == 1 and value.stride(2) == D and ts >= M * D and ts % 4 == 0
            and (B == 1 or value.stride(0) == S * ts) and value.data_ptr() % 16 == 0,
            "value must be [B, S, M, D] float32, dense or a 16-byte aligned column block of a [B, S, C] tensor")
    mask = None
    if value_mask is not None:
        _assert(value_mask.shape == (B, S) and value_mask.dtype == torch.bool and value_mask.device == value.device,
                "value_mask must be a bool [B, S] tensor on the value's device")
        mask = value_mask.contiguous()
    return ts, (mask.data_ptr() if mask is not None else None), mask


def _fused_forward_view(value, spatial_shapes, level_start_index, proj, reference_points, value_mask, save, name):
    B, S, M, D = value.shape
    Lq = proj.shape[1]
    L = P = 4
    _assert(proj.is_contiguous() and proj.shape[2] == M * 48 and proj.dtype == torch.float32, "proj must be [B, Lq, M*48] float32")
    geom = host_geometry(spatial_shapes, level_start_index)
    ts, mask_ptr, mask = _value_view(value, value_mask)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    loc = torch.empty((B, M, L, Lq, P, 2), dtype=value.dtype, device=value.device) if save else None
    attw = torch.empty((B, M, L, Lq, P), dtype=value.dtype, device=value.device) if save else None
    timer = LaunchTimer.active
    with on_device(value.device):
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("fwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        code = _lib.load().msda_fused_forward_view_f32(
            value.data_ptr(), ts, mask_ptr, spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(),
            proj.data_ptr() + M * 32 * 4, reference_points.data_ptr(), reference_points.size(3), out.data_ptr(),
            loc.data_ptr() if save else None, attw.data_ptr() if save else None, B, S, M, D, L, Lq, P, M * 48, M * 48,
            geom[0], geom[1], raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, name)
    return out, loc, attw


# Training: the saved backward's plan on a side stream right behind the forward (ABI v9).  OPT-IN (MONOSOWA_PLAN_AHEAD=1): measured
# in same-box A/B at B = 16 it takes 32 us off every encoder backward (0.907 -> 0.875 ms: the three plan kernels leave the critical
# path) and ADDS 1.1 ms to the step (69.1 -> 70.2 ms): three cross-stream dependencies per step (a completion signal behind each
# forward kernel, the allocator's stream bookkeeping for the workspace) cost the main queue more than the 0.1 ms they hide.
PLAN_AHEAD = os.environ.get("MONOSOWA_PLAN_AHEAD", "0") != "0"
_PLAN_STREAMS = {}


def plan_saved_backward(value, spatial_shapes, level_start_index, loc):
    """Starts what ``ms_deform_attn_fused_backward_merged_saved`` would launch in front of its scatter -- directional statistics,
    per-head plan, candidate tables: three small dependent kernels that need the forward's saved locations ``loc`` and nothing else
    -- on a side stream, behind the forward kernel that wrote ``loc`` and beside whatever the main stream runs next.  Returns a
    handle for the backward's ``plan=`` argument (workspace + event), or None when the call would not use a directional plan."""
    if not PLAN_AHEAD or not loc.is_cuda:
        return None
    B, S, M, D = value.shape
    L = P = 4
    Lq = loc.shape[3]
    lib = _lib.load()
    geom = host_geometry(spatial_shapes, level_start_index)
    ws_bytes = lib.msda_backward_workspace_bytes(B, S, M, D, L, Lq, P, 4)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device)
    main = torch.cuda.current_stream(value.device)
    side = _PLAN_STREAMS.get(value.device.index)
    if side is None:
        side = _PLAN_STREAMS[value.device.index] = torch.cuda.Stream(value.device)
    side.wait_stream(main)                                   # behind the forward kernel (and the allocation of `ws`)
    with torch.cuda.stream(side), on_device(value.device):
        code = lib.msda_saved_plan_f32(loc.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), value.stride(1), B, S, M,
                                       D, L, Lq, P, M * 48, M * 48, geom[0], geom[1], ws.data_ptr(), ws_bytes, _lib.raw_stream())
        if code == 0:
            event = side.record_event()
    if code != 0:
        return None                                          # (MSDA_E_UNSUPPORTED: the backward plans for itself)
    ws.record_stream(side)                                   # freed early (no backward)? the allocator waits for the side stream
    loc.record_stream(side)
    return ws, event, lib.msda_options_stamp()               # (the stamp: a plan is only good under the options it was made with)


def _fused_backward_view(value, spatial_shapes, level_start_index, a, b, saved, reference_points, grad_output, value_mask, name, plan=None):
    B, S, M, D = value.shape
    L = P = 4
    Lq = a.shape[3] if saved else a.shape[1]
    _assert(grad_output.is_contiguous() and grad_output.numel() == B * Lq * M * D, "grad_output shape mismatch")
    lib = _lib.load()
    geom = host_geometry(spatial_shapes, level_start_index)
    ts, mask_ptr, mask = _value_view(value, value_mask)
    grad_value = torch.empty((B, S, M, D), dtype=value.dtype, device=value.device)         # always dense
    grad_proj = torch.empty((B, Lq, M * 48), dtype=value.dtype, device=value.device)
    ws_bytes = lib.msda_backward_workspace_bytes(B, S, M, D, L, Lq, P, 4)
    planned = saved and plan is not None and plan[0].numel() == ws_bytes and plan[0].device == value.device and \
        (len(plan) < 3 or plan[2] == lib.msda_options_stamp())          # options changed since plan time: plan afresh
    ws = plan[0] if planned else torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device)
    timer = LaunchTimer.active
    with on_device(value.device):
        if planned:
            torch.cuda.current_stream().wait_event(plan[1])       # the plan kernels have run (side stream)
        stream = torch.cuda.current_stream() if timer is not None else None      # (a Stream object only for the events)
        raw = _lib.raw_stream()
        if timer is not None:
            e0, e1 = timer.bracket("bwd", (B, S, M, D, L, Lq, P))
            e0.record(stream)
        if planned:
            code = lib.msda_fused_backward_view_planned_f32(
                value.data_ptr(), ts, mask_ptr, spatial_shapes.data_ptr(), level_start_index.data_ptr(), a.data_ptr(), b.data_ptr(),
                reference_points.data_ptr(), reference_points.size(3), grad_output.data_ptr(), grad_value.data_ptr(),
                grad_proj.data_ptr(), grad_proj.data_ptr() + M * 32 * 4, B, S, M, D, L, Lq, P, M * 48, M * 48, geom[0], geom[1],
                ws.data_ptr(), ws_bytes, raw)
            if code == -3:                                    # MSDA_E_UNSUPPORTED: this call does not run on a directional plan after all
                planned = False
        if not planned:
            code = lib.msda_fused_backward_view_f32(
                value.data_ptr(), ts, mask_ptr, spatial_shapes.data_ptr(), level_start_index.data_ptr(), a.data_ptr(), b.data_ptr(),
                1 if saved else 0, reference_points.data_ptr(), reference_points.size(3), grad_output.data_ptr(), grad_value.data_ptr(),
                grad_proj.data_ptr(), grad_proj.data_ptr() + M * 32 * 4, B, S, M, D, L, Lq, P, M * 48, M * 48, geom[0], geom[1],
                ws.data_ptr(), ws_bytes, raw)
        if timer is not None:
            e1.record(stream)
    _lib.check(code, name)
    return grad_value, grad_proj


def ms_deform_attn_fused_forward_merged(value, spatial_shapes, level_start_index, proj, reference_points, value_mask=None):
    """Fused operator on ONE projection output ``proj`` [B, Lq, M*48] = (offsets M*32 | logits M*16) of a merged
    sampling_offsets / attention_weights GEMM, read in place through row strides.  ``value`` may be a column block of a
    wider projection (token stride), ``value_mask`` [B, S] bool marks padded tokens whose value rows count as zero
    (msda_fused_forward_view_f32, ABI v7)."""
    return _fused_forward_view(value, spatial_shapes, level_start_index, proj, reference_points, value_mask, False,
                               "ms_deform_attn_fused_forward_merged")[0]


def ms_deform_attn_fused_backward_merged(value, spatial_shapes, level_start_index, proj, reference_points, grad_output,
                                         value_mask=None):
    """-> grad_value (dense [B, S, M, D]; zero rows for padded tokens), grad_proj [B, Lq, M*48] (grad offsets | grad logits in
    the layout of ``proj``)."""
    _assert(proj.is_contiguous() and proj.shape[2] == value.shape[2] * 48, "proj must be [B, Lq, M*48]")
    return _fused_backward_view(value, spatial_shapes, level_start_index, proj, proj[:, :, value.shape[2] * 32:], False,
                                reference_points, grad_output, value_mask, "ms_deform_attn_fused_backward_merged")


def fused_save_supported(value, spatial_shapes, level_start_index, Lq, ref_dim=2):
    """True when the training pair msda_fused_forward_save_f32 / msda_fused_backward_saved_f32 (ABI v6) covers this call:
    the self-attention shape (Lq == S) on the window / row-tile kernels."""
    B, S, M, D = value.shape
    geom = host_geometry(spatial_shapes, level_start_index)
    # ``value`` may be a column block of a wider projection: its token stride bounds the kernels' plane addressing too
    return bool(_lib.load().msda_fused_save_supported_view(S, M, D, 4, Lq, 4, ref_dim, value.stride(1), M * 48, M * 48,
                                                           geom[0], geom[1]))


def ms_deform_attn_fused_forward_merged_save(value, spatial_shapes, level_start_index, proj, reference_points, value_mask=None):
    """As ``ms_deform_attn_fused_forward_merged``; also returns the sampling locations [B, M, 4, Lq, 4, 2] and attention
    weights [B, M, 4, Lq, 4] (level-major) the kernel evaluated, for ``ms_deform_attn_fused_backward_merged_saved``."""
    return _fused_forward_view(value, spatial_shapes, level_start_index, proj, reference_points, value_mask, True,
                               "ms_deform_attn_fused_forward_merged_save")


def ms_deform_attn_fused_backward_merged_saved(value, spatial_shapes, level_start_index, loc, attw, reference_points,
                                               grad_output, value_mask=None, plan=None):
    """-> grad_value, grad_proj [B, Lq, M*48] (grad offsets | grad logits), from the saved locations / weights.
    ``plan``: the handle ``plan_saved_backward`` returned for these ``loc`` (or None: the backward plans for itself)."""
    return _fused_backward_view(value, spatial_shapes, level_start_index, loc, attw, True, reference_points, grad_output,
                                value_mask, "ms_deform_attn_fused_backward_merged_saved", plan=plan)
