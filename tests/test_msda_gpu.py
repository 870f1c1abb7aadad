"""GPU parity tests of the MSDA HIP kernels, called through the C-ABI
(monosowa_amd.MultiScaleDeformableAttention -> libmonosowa_msda.so).

Checker: oracle/ (C restatement pinned to the reference's Python by test_oracle_golden.py) and
the committed golden vectors.  Tolerances: f64 1e-9 rel; f32 1e-4 rel of the tensor's max
(north_star: "<=1e-4 rel fp32"); grad_value in f32 also carries atomic-order noise.
Mirrors the reference's ops/test.py: fwd vs PyTorch in double and float (:32-60), gradcheck in
double for D in {30,32,64,71,1025} (:63-86).
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import msda_oracle as O

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "msda_*.npz")))
KITTI_LEVELS = [(48, 160), (24, 80), (12, 40), (6, 20)]          # 1280x384, strides 8..64 -> S=10200


def _msda():
    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    return MSDA


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, want, rel, what):
    got = got.detach().cpu().numpy().reshape(want.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max()) / scale
    assert err <= rel, "%s: max err / max|ref| = %.3e > %.1e" % (what, err, rel)


def _oracle_want(value, shapes, lsi, loc, w, go):
    """Expected results.  f64 inputs: the f64 oracle.  f32 inputs: the f32 oracle (same arithmetic,
    hence the same floor()/in-bounds decisions as the kernel -- grad_loc jumps across pixel borders, so
    it cannot be judged against f64 arithmetic) plus, for the continuous outputs (out, grad_value,
    grad_attw), the f64 oracle on the same f32-rounded inputs."""
    want = (O.forward(value, shapes, lsi, loc, w),) + O.backward(value, shapes, lsi, loc, w, go)
    if value.dtype == np.float64:
        return want, None
    d = lambda a: a.astype(np.float64)
    w64 = (O.forward(d(value), shapes, lsi, d(loc), d(w)),) + O.backward(d(value), shapes, lsi, d(loc), d(w), d(go))
    return want, w64


def _close_elementwise(got, want, rel, what):
    """Element-wise check for `out` (a convex combination of value rows: no cancellation beyond the sampled rows'
    own magnitude): |err| <= rel * (|ref| + rms(ref)) for every element."""
    got = got.detach().cpu().numpy().reshape(want.shape).astype(np.float64)
    want = want.astype(np.float64)
    floor = float(np.sqrt((want ** 2).mean())) + 1e-30
    err = float((np.abs(got - want) / (np.abs(want) + floor)).max())
    assert err <= rel, "%s: max element-wise err = %.3e > %.1e" % (what, err, rel)


def _run_case(value, shapes, lsi, loc, attw, grad_out, want, rel, want64=None):
    """Forward through BOTH d32 variants: without a host pyramid (fwd_d32_kernel; the forward never synchronises to
    fetch one) and with it attached (the record / LDS-staged / windowed production kernels bench.py and training run)."""
    MSDA = _msda()
    v, s, i, lc, w, go = map(_dev, (value, shapes, lsi, loc, attw, grad_out))
    out_plain = MSDA.ms_deform_attn_forward(v, s, i, lc, w, 64)
    assert tuple(out_plain.shape) == (value.shape[0], loc.shape[1], value.shape[2] * value.shape[3])
    _close(out_plain, want[0], rel, "out (no host pyramid)")
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    out = MSDA.ms_deform_attn_forward(v, s, i, lc, w, 64)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, s, i, lc, w, go, 64)
    torch.cuda.synchronize()
    _close(out, want[0], rel, "out")
    _close_elementwise(out, want[0], 10 * rel, "out")
    _close(gv, want[1], rel, "grad_value")
    _close(gl, want[2], rel, "grad_loc")
    _close(gw, want[3], rel, "grad_attw")
    if want64 is not None:
        _close(out, want64[0], rel, "out vs f64")
        _close(gv, want64[1], rel, "grad_value vs f64")
        _close(gw, want64[3], rel, "grad_attw vs f64")


@pytest.mark.parametrize("case", CASES)
def test_golden_vectors_from_reference(case, golden_dir):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    rel = 1e-9 if g["value"].dtype == np.float64 else 1e-4
    _run_case(g["value"], g["shapes"], g["lsi"], g["loc"], g["attw"], g["grad_out"],
              (g["out"], g["grad_value"], g["grad_loc"], g["grad_attw"]), rel)


def _random_case(seed, B, M, D, Lq, levels, P, dtype, loc_lo=-0.3, loc_hi=1.3):
    rng = np.random.default_rng(seed)
    shapes = np.array(levels, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    L = len(levels)
    value = rng.standard_normal((B, S, M, D)).astype(dtype)
    loc = rng.uniform(loc_lo, loc_hi, (B, Lq, M, L, P, 2)).astype(dtype)
    w = rng.uniform(0, 1, (B, Lq, M, L, P))
    w = (w / w.sum((-1, -2), keepdims=True)).astype(dtype)
    go = rng.standard_normal((B, Lq, M * D)).astype(dtype)
    return value, shapes, lsi, loc, w, go


@pytest.mark.parametrize("B,M,D,Lq,levels,P,dtype", [
    (2, 8, 32, 550, [(12, 40), (6, 20), (3, 10), (2, 5)], 4, np.float32),     # decoder, train
    (3, 8, 32, 50, [(12, 40), (6, 20), (3, 10), (2, 5)], 4, np.float32),      # decoder, eval
    (1, 8, 32, 1275, [(24, 40), (12, 20), (6, 10), (3, 5)], 4, np.float32),   # encoder-like Lq == S
    (2, 8, 32, 77, [(12, 40), (6, 20), (3, 10), (2, 5)], 4, np.float64),      # generic path, D=32
    (1, 2, 30, 9, [(6, 4), (3, 2)], 2, np.float64),
    (1, 2, 64, 9, [(6, 4), (3, 2)], 2, np.float32),
    (1, 2, 71, 9, [(6, 4), (3, 2)], 2, np.float64),
    (1, 2, 1025, 3, [(6, 4), (3, 2)], 2, np.float64),
    (2, 3, 5, 4, [(1, 1), (2, 7), (5, 1)], 3, np.float32),                     # ragged levels
    (1, 8, 32, 33, [(7, 9), (4, 5), (2, 3), (1, 2)], 4, np.float32),           # d32 path, odd pair count
    (1, 4, 32, 5, [(7, 9), (4, 5), (2, 3), (1, 2), (1, 1)], 2, np.float32),    # D=32 but L*P != 16 -> generic
])
def test_random_vs_c_oracle(B, M, D, Lq, levels, P, dtype):
    value, shapes, lsi, loc, w, go = _random_case(B * 131 + D, B, M, D, Lq, levels, P, dtype)
    want, want64 = _oracle_want(value, shapes, lsi, loc, w, go)
    _run_case(value, shapes, lsi, loc, w, go, want, 1e-9 if dtype == np.float64 else 1e-4, want64)


def test_kitti_geometry_encoder_one_sample_vs_c_oracle():
    """Full 1280x384 geometry (S = Lq = 10200, M=8, D=32, L=4, P=4), B=2, locations = pixel-centre
    reference grid + up to 4 px offsets (what the encoder produces at init, ms_deform_attn.py:106-114)."""
    rng = np.random.default_rng(7)
    shapes = np.array(KITTI_LEVELS, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    assert S == 10200
    B, M, D, L, P = 2, 8, 32, 4, 4
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                          for h, w in KITTI_LEVELS])                                   # [S,2] (x,y)
    off = rng.uniform(-4, 4, (B, S, M, L, P, 2)) / shapes[None, None, None, :, None, ::-1]
    loc = (ref[None, :, None, None, None, :] + off).astype(np.float32)
    value = rng.standard_normal((B, S, M, D)).astype(np.float32)
    w = rng.standard_normal((B, S, M, L * P))
    w = np.exp(w) / np.exp(w).sum(-1, keepdims=True)
    w = w.reshape(B, S, M, L, P).astype(np.float32)
    go = rng.standard_normal((B, S, M * D)).astype(np.float32)
    want, want64 = _oracle_want(value, shapes, lsi, loc, w, go)
    _run_case(value, shapes, lsi, loc, w, go, want, 1e-4, want64)


def _kitti_encoder_inputs(B, seed, offset_px=4.0):
    rng = np.random.default_rng(seed)
    shapes = np.array(KITTI_LEVELS, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    M, D, L, P = 8, 32, 4, 4
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                          for h, w in KITTI_LEVELS]).astype(np.float32)                # [S,2] (x,y), pixel centres
    offsets = rng.uniform(-offset_px, offset_px, (B, S, M, L, P, 2)).astype(np.float32)
    logits = rng.standard_normal((B, S, M, L * P)).astype(np.float32)
    value = rng.standard_normal((B, S, M, D)).astype(np.float32)
    go = rng.standard_normal((B, S, M * D)).astype(np.float32)
    return shapes, lsi, ref, offsets, logits, value, go


@pytest.mark.parametrize("B,check,offset_px", [(2, (0, 1), 4.0), (16, (3, 15), 4.0), (2, (0, 1), 11.0)])
def test_fused_strided_operator_at_the_kitti_pyramid_vs_c_oracle(B, check, offset_px):
    """The operator the train step runs (msda_fused_*_strided_f32 on a merged [B, S, 384] projection, host pyramid
    attached, 48x160 / 24x80 / 12x40 / 6x20) against the C oracle fed with the PyTorch-evaluated prologue
    (softmax, ref + offset / (W, H): ms_deform_attn.py:146-152).  B = 16 is BASELINE configs[1]; the oracle checks two
    of its samples.  offset_px = 11: most taps leave any tile-local window (fallback paths)."""
    MSDA = _msda()
    shapes, lsi, ref, offsets, logits, value, go = _kitti_encoder_inputs(B, 23 + B, offset_px)
    S, M = value.shape[1], value.shape[2]
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    proj = torch.cat([_dev(offsets).reshape(B, S, M * 32), _dev(logits).reshape(B, S, M * 16)], -1).contiguous()
    refp = _dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
    v, g = _dev(value), _dev(go)
    out = MSDA.ms_deform_attn_fused_forward_merged(v, s, i, proj, refp)
    gv, gproj = MSDA.ms_deform_attn_fused_backward_merged(v, s, i, proj, refp, g)
    torch.cuda.synchronize()
    # prologue in PyTorch on the device (the same f32 expressions the module evaluates), core through the oracle
    off_t = _dev(offsets).requires_grad_(True)
    log_t = _dev(logits).requires_grad_(True)
    norm = torch.stack([s[:, 1], s[:, 0]], -1).float()
    loc_t = refp[:, :, None, :, None, :] + off_t / norm[None, None, None, :, None, :]
    aw_t = torch.softmax(log_t, -1).view(B, S, M, 4, 4)
    for b in check:
        loc_b, aw_b = loc_t[b:b + 1].detach().cpu().numpy(), aw_t[b:b + 1].detach().cpu().numpy()
        want = (O.forward(value[b:b + 1], shapes, lsi, loc_b, aw_b),) + O.backward(value[b:b + 1], shapes, lsi, loc_b, aw_b, go[b:b + 1])
        _close(out[b:b + 1], want[0], 1e-4, "fused out[%d]" % b)
        _close_elementwise(out[b:b + 1], want[0], 1e-3, "fused out[%d]" % b)
        _close(gv[b:b + 1], want[1], 1e-4, "fused grad_value[%d]" % b)
        # chain the oracle's grad_loc / grad_attw through the prologue with autograd
        gl = torch.from_numpy(want[2]).cuda()
        ga = torch.from_numpy(want[3]).cuda()
        g_off, g_log = torch.autograd.grad([loc_t[b:b + 1], aw_t[b:b + 1]], [off_t, log_t], [gl, ga], retain_graph=True)
        got_off = gproj[b, :, :M * 32].reshape(1, S, M, 4, 4, 2)
        got_log = gproj[b, :, M * 32:].reshape(1, S, M, 16)
        assert (got_off - g_off[b:b + 1]).abs().max() <= 1e-4 * g_off[b].abs().max(), "grad_offsets[%d]" % b
        assert (got_log - g_log[b:b + 1]).abs().max() <= 1e-4 * g_log[b].abs().max(), "grad_logits[%d]" % b


@pytest.mark.parametrize("option,value", [("scatter_lists", 1), ("directional", 0), ("plan_fused", 0)])
def test_saved_backward_alternative_scan_sources_compute_the_same_gradients(option, value):
    """The saved backward's two opt-in / fallback scan sources -- exact scan lists binned from the saved locations
    (msda_set_option("scatter_lists", 1): msda_bin.hip) and the isotropic host plan ("directional", 0) -- against the default
    (geometric scan behind the device-side directional plan, made by ONE fused launch), offsets of 6.5 px so that window-outside and
    far points occur.  ("plan_fused", 0): the plan from the three separate kernels (larger statistics sample: the bounds may differ
    by a pixel, the gradients may not)."""
    MSDA = _msda()
    from monosowa_amd import _lib
    B = 2
    shapes, lsi, ref, offsets, logits, value_, go = _kitti_encoder_inputs(B, 78, 6.5)
    S, M = value_.shape[1], value_.shape[2]
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    proj = torch.cat([_dev(offsets).reshape(B, S, M * 32), _dev(logits).reshape(B, S, M * 16)], -1).contiguous()
    refp = _dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
    v, g = _dev(value_), _dev(go)
    _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(v, s, i, proj, refp)
    gv_a, gp_a = MSDA.ms_deform_attn_fused_backward_merged_saved(v, s, i, loc, attw, refp, g)
    lib = _lib.load()
    default = {"scatter_lists": 0, "directional": 1, "plan_fused": 1}[option]
    assert lib.msda_set_option(option.encode(), value) == 0
    try:
        gv_b, gp_b = MSDA.ms_deform_attn_fused_backward_merged_saved(v, s, i, loc, attw, refp, g)
        torch.cuda.synchronize()
    finally:
        assert lib.msda_set_option(option.encode(), default) == 0
    for name, a, b in (("grad_value", gv_a, gv_b), ("grad_proj", gp_a, gp_b)):
        assert (a - b).abs().max() <= 2e-6 * a.abs().max(), (name, ((a - b).abs().max() / a.abs().max()).item())


def test_saved_prologue_backward_equals_the_recomputing_backward():
    """ABI v6: msda_fused_forward_save_f32 + msda_fused_backward_saved_f32 (the backward reads the sampling locations /
    attention weights the forward stored) against the v5 pair that re-evaluates the prologue -- same output bit for bit,
    gradients to 2e-6 of their max; the saved tensors against the PyTorch-evaluated prologue."""
    MSDA = _msda()
    B = 2
    shapes, lsi, ref, offsets, logits, value, go = _kitti_encoder_inputs(B, 77, 6.5)        # some far / out-of-window points too
    S, M = value.shape[1], value.shape[2]
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    proj = torch.cat([_dev(offsets).reshape(B, S, M * 32), _dev(logits).reshape(B, S, M * 16)], -1).contiguous()
    refp = _dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
    v, g = _dev(value), _dev(go)
    assert MSDA.fused_save_supported(v, s, i, S)
    out_a = MSDA.ms_deform_attn_fused_forward_merged(v, s, i, proj, refp)
    gv_a, gp_a = MSDA.ms_deform_attn_fused_backward_merged(v, s, i, proj, refp, g)
    out_b, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(v, s, i, proj, refp)
    gv_b, gp_b = MSDA.ms_deform_attn_fused_backward_merged_saved(v, s, i, loc, attw, refp, g)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    norm = torch.stack([s[:, 1], s[:, 0]], -1).float()
    loc_t = refp[:, :, None, :, None, :] + _dev(offsets) / norm[None, None, None, :, None, :]
    aw_t = torch.softmax(_dev(logits), -1).view(B, S, M, 4, 4)
    # saved layout: level-major [B, M, L, Lq, P(, 2)]
    assert (loc.permute(0, 3, 1, 2, 4, 5) - loc_t).abs().max() <= 1e-6 and (attw.permute(0, 3, 1, 2, 4) - aw_t).abs().max() <= 1e-6
    for name, a, b in (("grad_value", gv_a, gv_b), ("grad_proj", gp_a, gp_b)):
        assert (a - b).abs().max() <= 2e-6 * a.abs().max(), (name, ((a - b).abs().max() / a.abs().max()).item())
    assert not MSDA.fused_save_supported(v, s, i, 550)                    # decoder shape: the v5 pair


def test_saved_operator_b16_with_trained_like_offsets_vs_c_oracle():
    """The training pair (msda_fused_forward_save / backward_saved: window gather, cell scatter, device-side directional plan) at
    BASELINE configs[1] (B = 16) with sampling offsets of N(0, 8 px) at every level -- far beyond the module's initial +-4 px:
    three quarters of the points leave any LDS window (pair-served global loads), half lie beyond the scatter's scan bounds
    (half-wave row atomics).  Reference behaviour: offsets are unbounded (ms_deform_attn.py:145-155).  One sample against the C
    oracle through the PyTorch-evaluated prologue."""
    MSDA = _msda()
    B = 16
    shapes, lsi, ref, offsets, logits, value, go = _kitti_encoder_inputs(B, 91)
    rng = np.random.default_rng(92)
    offsets = (8.0 * rng.standard_normal(offsets.shape)).astype(np.float32)
    S, M = value.shape[1], value.shape[2]
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    proj = torch.cat([_dev(offsets).reshape(B, S, M * 32), _dev(logits).reshape(B, S, M * 16)], -1).contiguous()
    refp = _dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
    v, g = _dev(value), _dev(go)
    assert MSDA.fused_save_supported(v, s, i, S)
    out, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(v, s, i, proj, refp)
    gv, gproj = MSDA.ms_deform_attn_fused_backward_merged_saved(v, s, i, loc, attw, refp, g)
    torch.cuda.synchronize()
    b = 11
    off_t = _dev(offsets[b:b + 1]).requires_grad_(True)
    log_t = _dev(logits[b:b + 1]).requires_grad_(True)
    norm = torch.stack([s[:, 1], s[:, 0]], -1).float()
    loc_t = refp[b:b + 1, :, None, :, None, :] + off_t / norm[None, None, None, :, None, :]
    aw_t = torch.softmax(log_t, -1).view(1, S, M, 4, 4)
    loc_b, aw_b = loc_t.detach().cpu().numpy(), aw_t.detach().cpu().numpy()
    want = (O.forward(value[b:b + 1], shapes, lsi, loc_b, aw_b),) + O.backward(value[b:b + 1], shapes, lsi, loc_b, aw_b, go[b:b + 1])
    _close(out[b:b + 1], want[0], 1e-4, "out")
    _close_elementwise(out[b:b + 1], want[0], 1e-3, "out")
    _close(gv[b:b + 1], want[1], 1e-4, "grad_value")
    g_off, g_log = torch.autograd.grad([loc_t, aw_t], [off_t, log_t], [torch.from_numpy(want[2]).cuda(), torch.from_numpy(want[3]).cuda()])
    got_off = gproj[b:b + 1, :, :M * 32].reshape(1, S, M, 4, 4, 2)
    got_log = gproj[b:b + 1, :, M * 32:].reshape(1, S, M, 16)
    assert (got_off - g_off).abs().max() <= 1e-4 * g_off.abs().max(), "grad_offsets"
    assert (got_log - g_log).abs().max() <= 1e-4 * g_log.abs().max(), "grad_logits"


def test_unfused_production_forward_b16_two_samples_vs_c_oracle():
    """ms_deform_attn_forward / _backward with the host pyramid attached at BASELINE configs[1] (B = 16)."""
    MSDA = _msda()
    B = 16
    shapes, lsi, ref, offsets, logits, value, go = _kitti_encoder_inputs(B, 41)
    loc = (ref[None, :, None, None, None, :] + offsets / shapes[None, None, None, :, None, ::-1]).astype(np.float32)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    aw = (e / e.sum(-1, keepdims=True)).reshape(B, -1, 8, 4, 4).astype(np.float32)
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, shapes.tolist(), lsi.tolist())
    v, lc, w, g = _dev(value), _dev(loc), _dev(aw), _dev(go)
    out = MSDA.ms_deform_attn_forward(v, s, i, lc, w, 64)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, s, i, lc, w, g, 64)
    torch.cuda.synchronize()
    for b in (0, 9):
        want = (O.forward(value[b:b + 1], shapes, lsi, loc[b:b + 1], aw[b:b + 1]),) + \
            O.backward(value[b:b + 1], shapes, lsi, loc[b:b + 1], aw[b:b + 1], go[b:b + 1])
        _close(out[b:b + 1], want[0], 1e-4, "out[%d]" % b)
        _close_elementwise(out[b:b + 1], want[0], 1e-3, "out[%d]" % b)
        _close(gv[b:b + 1], want[1], 1e-4, "grad_value[%d]" % b)
        _close(gl[b:b + 1], want[2], 1e-4, "grad_loc[%d]" % b)
        _close(gw[b:b + 1], want[3], 1e-4, "grad_attw[%d]" % b)


def test_host_pyramid_that_contradicts_the_tensors_is_refused():
    """A host pyramid whose levels do not tile [0, S) must come back as MSDA_E_SHAPE, not as an out-of-bounds kernel."""
    MSDA = _msda()
    from monosowa_amd import _lib
    levels = [(12, 40), (6, 20), (3, 10), (2, 5)]
    value, shapes, lsi, loc, w, go = _random_case(5, 1, 8, 32, 64, levels, 4, np.float32)
    v, s, i, lc, ww, g = map(_dev, (value, shapes, lsi, loc, w, go))
    bad_shapes = [(12, 40), (6, 20), (3, 10), (4, 5)]                     # 10 rows more than S
    geom = MSDA.attach_host_geometry(s, i, bad_shapes, lsi.tolist())
    with pytest.raises(RuntimeError, match="dimension"):
        MSDA.ms_deform_attn_forward(v, s, i, lc, ww, 64, host_geom=geom)
    with pytest.raises(RuntimeError, match="dimension"):
        MSDA.ms_deform_attn_backward(v, s, i, lc, ww, g, 64, host_geom=geom)
    geom = MSDA.attach_host_geometry(s, i, levels, [0, 480, 600, 999])     # a level start past the end
    with pytest.raises(RuntimeError, match="dimension"):
        MSDA.ms_deform_attn_forward(v, s, i, lc, ww, 64, host_geom=geom)
    del _lib


def test_full_batch_properties_b16():
    """BASELINE configs[1] size (B=16, S=Lq=10200): size-independent properties.
    (1) a constant value field sampled strictly inside gives out = c * sum(w) = c;
    (2) linearity in value; (3) f32 d32 kernel agrees with the f64 generic kernel."""
    MSDA = _msda()
    torch.manual_seed(0)
    dev = "cuda"
    shapes = torch.tensor(KITTI_LEVELS, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    B, S, M, D, L, P = 16, 10200, 8, 32, 4, 4
    loc = torch.rand(B, S, M, L, P, 2, device=dev) * 0.8 + 0.1       # footprint fully inside every level
    w = torch.softmax(torch.randn(B, S, M, L * P, device=dev), -1).view(B, S, M, L, P)
    const = torch.full((B, S, M, D), 1.5, device=dev)
    out = MSDA.ms_deform_attn_forward(const, shapes, lsi, loc, w, 64)
    assert torch.allclose(out, torch.full_like(out, 1.5), rtol=0, atol=2e-6)
    v1 = torch.randn(B, S, M, D, device=dev)
    v2 = torch.randn(B, S, M, D, device=dev)
    loc = torch.rand(B, S, M, L, P, 2, device=dev) * 1.4 - 0.2
    o1 = MSDA.ms_deform_attn_forward(v1, shapes, lsi, loc, w, 64)
    o2 = MSDA.ms_deform_attn_forward(v2, shapes, lsi, loc, w, 64)
    o12 = MSDA.ms_deform_attn_forward(v1 + 2 * v2, shapes, lsi, loc, w, 64)
    assert (o12 - (o1 + 2 * o2)).abs().max() <= 1e-4 * o12.abs().max()
    o64 = MSDA.ms_deform_attn_forward(v1[:2].double(), shapes, lsi, loc[:2].double(), w[:2].double(), 64)
    assert (o1[:2].double() - o64).abs().max() <= 1e-4 * o64.abs().max()
    # backward: sum over all grad_value equals sum over (q,m,c) of grad_out * (sum of in-bounds tap weights)
    go = torch.randn(B, S, M * D, device=dev)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v1, shapes, lsi, loc, w, go, 64)
    gv64, gl64, gw64 = MSDA.ms_deform_attn_backward(v1[:2].double(), shapes, lsi, loc[:2].double(), w[:2].double(), go[:2].double(), 64)
    for a, b, n in ((gv[:2], gv64, "grad_value"), (gw[:2], gw64, "grad_attw")):
        assert (a.double() - b).abs().max() <= 1e-4 * b.abs().max(), n
    # grad_loc jumps across pixel borders, where f32 and f64 floor() may differ: allow a handful of points
    bad = ((gl[:2].double() - gl64).abs() > 1e-4 * gl64.abs().max()).sum().item()
    assert bad <= 1e-5 * gl64.numel(), bad
    # <grad_out, J v> == <J^T grad_out, v>  (adjoint identity of the value path)
    lhs = (go.double() * o1.double()).sum()
    rhs = (gv.double() * v1.double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


def test_gradcheck_double_like_reference():
    """ops/test.py:63-86 -- numerical gradient check in double through the autograd Function."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    for D in (30, 32, 64, 71):
        value = (torch.rand(N, S, M, D).cuda() * 0.01).double().requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2).cuda().double().requires_grad_(True)
        aw = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
        assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, aw, 2))


@pytest.mark.parametrize("D", [1025, 2048, 3096])
def test_gradcheck_double_large_channel_counts(D):
    """The rest of ops/test.py:85's channel list (the reference's multi-block / global-memory backward variants,
    cuh:731-920).  The full Jacobian of the value path is 8 GB at D = 2048, so these use gradcheck's fast mode
    (directional derivatives along random vectors) -- same tolerance, every input covered."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, D).cuda() * 0.01).double().requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2).cuda().double().requires_grad_(True)
    aw = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, lsi, loc, aw, 2), fast_mode=True, nondet_tol=1e-8)


def test_preconditions_on_gpu():
    MSDA = _msda()
    shapes = torch.tensor([[2, 2]], dtype=torch.long).cuda()
    lsi = torch.zeros(1, dtype=torch.long).cuda()
    v = torch.zeros(3, 4, 1, 4).cuda()
    loc = torch.zeros(3, 1, 1, 1, 1, 2).cuda()
    w = torch.zeros(3, 1, 1, 1, 1).cuda()
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, w, 2)          # 3 % 2 != 0 (cu:50-52)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(torch.zeros(3, 4, 2, 4).cuda()[:, :, :1], shapes, lsi, loc, w, 64)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        MSDA.ms_deform_attn_forward(v, shapes.cpu(), lsi, loc, w, 64)
    out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, w, 3)
    assert out.shape == (3, 1, 4) and not out.any()
    # no queries at all: the reference hands back its empty output / untouched zero gradients (cu:54, 121-123)
    loc0, w0 = torch.zeros(3, 0, 1, 1, 1, 2).cuda(), torch.zeros(3, 0, 1, 1, 1).cuda()
    out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc0, w0, 3)
    assert out.shape == (3, 0, 4)
    gv, gl, gw = MSDA.ms_deform_attn_backward(torch.ones_like(v), shapes, lsi, loc0, w0, torch.zeros(3, 0, 4).cuda(), 3)
    assert gv.shape == v.shape and not gv.any() and gl.shape == loc0.shape and gw.shape == w0.shape


def test_all_points_outside_and_nan_locations():
    MSDA = _msda()
    shapes = torch.tensor(KITTI_LEVELS, dtype=torch.long).cuda()
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    v = torch.randn(1, 10200, 8, 32).cuda()
    loc = torch.full((1, 5, 8, 4, 4, 2), 7.0).cuda()
    loc[0, 1] = float("nan")
    loc[0, 2] = 1e30
    loc[0, 3] = -1e30
    w = torch.full((1, 5, 8, 4, 4), 1 / 16).cuda()
    out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, w, 64)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, w, torch.ones(1, 5, 256).cuda(), 64)
    assert not out.any() and not gv.any() and not gl.any() and not gw.any()


def test_token_linear_split_k_backward_matches_linear():
    from monosowa_amd.token_linear import token_linear
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 128).cuda()
    x = torch.randn(4, 10200, 256, device="cuda", requires_grad=True)
    go = torch.randn(4, 10200, 128, device="cuda")
    y = token_linear(x, lin)
    assert y.grad_fn is not None and "TokenLinear" in type(y.grad_fn).__name__
    y.backward(go)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    lin(x).backward(go)
    for a, b in zip(got, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max()
    small = torch.randn(2, 50, 256, device="cuda", requires_grad=True)
    assert "TokenLinear" not in type(token_linear(small, lin).grad_fn).__name__      # few rows: plain nn.Linear


def test_eval_forward_replays_from_a_hipgraph():
    """The C-ABI launches are capture-safe (no allocation, no host sync): the whole eval forward, MSDA included,
    replays from a hipGraph and reproduces the eager outputs."""
    import yaml
    from monosowa_amd.helpers.model_helper import build_model, to_mi355x_layout
    from monosowa_amd.helpers.tester_helper import GraphedForward
    from monosowa_amd.synthetic import make_batch
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "monodetr.yaml")))
    torch.manual_seed(1)
    model, _ = build_model(dict(cfg["model"], device="cuda"))
    model = to_mi355x_layout(model.cuda()).eval()
    inputs, calibs, targets, _ = make_batch(2, "cuda", seed=3, resolution=(320, 96))
    with torch.no_grad():
        eager = model(inputs, calibs, None, targets["img_size"])
    g = GraphedForward(model, inputs, calibs, targets["img_size"])
    inputs2, calibs2, targets2, _ = make_batch(2, "cuda", seed=4, resolution=(320, 96))
    out = {k: v.clone() for k, v in g(inputs, calibs, targets["img_size"]).items() if torch.is_tensor(v)}
    for k in ("pred_logits", "pred_boxes", "pred_depth", "pred_angle", "pred_3d_dim"):
        assert torch.allclose(out[k], eager[k], rtol=1e-4, atol=1e-5), k
    with torch.no_grad():
        eager2 = model(inputs2, calibs2, None, targets2["img_size"])
    out2 = g(inputs2, calibs2, targets2["img_size"])
    assert torch.allclose(out2["pred_boxes"], eager2["pred_boxes"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("gather,fixed,sorted_", [(0, 0, 0), (1, 1, 0), (2, 0, 0), (2, 1, 0), (2, 1, 2), (0, 0, 2)])
def test_every_kernel_generation_computes_the_same_function(gather, fixed, sorted_):
    """The earlier kernel generations stay selectable (msda_set_option) for A/B measurements; all of them must
    pass the same parity bar on the shipped geometry (encoder-like and decoder-like query sets)."""
    from monosowa_amd import _lib
    try:
        _lib.set_option("gather", gather)
        _lib.set_option("scatter_fixed", fixed)
        _lib.set_option("scatter_sorted", sorted_)
        for B, Lq, levels in ((2, 1275, [(24, 40), (12, 20), (6, 10), (3, 5)]), (2, 550, [(12, 40), (6, 20), (3, 10), (2, 5)])):
            value, shapes, lsi, loc, w, go = _random_case(gather * 7 + fixed, B, 8, 32, Lq, levels, 4, np.float32)
            want, want64 = _oracle_want(value, shapes, lsi, loc, w, go)
            MSDA = _msda()
            s = _dev(shapes)
            starts = [0]
            for h, wd in levels[:-1]:
                starts.append(starts[-1] + h * wd)
            MSDA.attach_host_geometry(s, _dev(lsi), levels, starts)           # lets the forward use the planned kernels
            v, i, lc, ww, g = map(_dev, (value, lsi, loc, w, go))
            out = MSDA.ms_deform_attn_forward(v, s, i, lc, ww, 64)
            gv, gl, gw = MSDA.ms_deform_attn_backward(v, s, i, lc, ww, g, 64)
            for got, ref, name in ((out, want[0], "out"), (gv, want[1], "grad_value"), (gl, want[2], "grad_loc"), (gw, want[3], "grad_attw")):
                _close(got, ref, 1e-4, "%s (gather=%d fixed=%d sorted=%d)" % (name, gather, fixed, sorted_))
            _close(gv, want64[1], 1e-4, "grad_value vs f64")
    finally:
        _lib.set_option("gather", 2)
        _lib.set_option("scatter_fixed", 1)
        _lib.set_option("scatter_sorted", 0)
    with pytest.raises(RuntimeError):
        _lib.set_option("no_such_option", 1)


def test_fixed_point_scatter_keeps_small_rows_accurate_under_outliers():
    """grad_out with one 1e6 outlier: the fixed-point accumulators are scaled per (batch, head) from max|grad_out|,
    rows far from the outlier must still match the f64 oracle to float precision relative to their own scale."""
    B, M, D, Lq, levels, P = 1, 8, 32, 300, [(12, 40), (6, 20), (3, 10), (2, 5)], 4
    value, shapes, lsi, loc, w, go = _random_case(77, B, M, D, Lq, levels, P, np.float32, 0.05, 0.95)
    go = go.reshape(B, Lq, M, D)
    go[0, 0, 0, :] = 1e6                                    # head 0 only
    go = go.reshape(B, Lq, M * D)
    d = lambda a: a.astype(np.float64)
    ref = O.backward(d(value), shapes, lsi, d(loc), d(w), d(go))[0]
    MSDA = _msda()
    gv = MSDA.ms_deform_attn_backward(*map(_dev, (value, shapes, lsi, loc, w, go)), 64)[0].cpu().numpy().astype(np.float64)
    other = ref[:, :, 1:, :]                                # heads without the outlier: own scale
    assert np.abs(gv[:, :, 1:, :] - other).max() <= 1e-5 * np.abs(other).max()
    assert np.abs(gv[:, :, 0, :] - ref[:, :, 0, :]).max() <= 1e-5 * np.abs(ref[:, :, 0, :]).max()


@pytest.mark.parametrize("ref_dim", [2, 6])
def test_fused_prologue_matches_the_unfused_operator(ref_dim):
    """msda_fused_*: softmax + sampling-location arithmetic inside the kernels (ms_deform_attn.py:146-155) against
    the same arithmetic in PyTorch followed by the unfused operator -- output and all three gradients."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedFunction
    MSDA = _msda()
    torch.manual_seed(ref_dim)
    levels = [(12, 40), (6, 20), (3, 10), (2, 5)]
    B, M, D, L, P, Lq = 2, 8, 32, 4, 4, 333
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    starts = [0]
    for h, w in levels[:-1]:
        starts.append(starts[-1] + h * w)
    MSDA.attach_host_geometry(shapes, lsi, levels, starts)
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, device="cuda", requires_grad=True)
    offsets = (torch.randn(B, Lq, M, L, P, 2, device="cuda") * 3).requires_grad_(True)
    logits = torch.randn(B, Lq, M, L * P, device="cuda", requires_grad=True)
    if ref_dim == 2:
        ref = torch.rand(B, Lq, L, 2, device="cuda") * 1.2 - 0.1
    else:
        ref = torch.cat([torch.rand(B, Lq, L, 2, device="cuda"), torch.rand(B, Lq, L, 4, device="cuda") * 0.3], -1)
    go = torch.randn(B, Lq, M * D, device="cuda")

    out_f = MSDeformAttnFusedFunction.apply(value, shapes, lsi, offsets, logits, ref)
    out_f.backward(go)
    got = [out_f.detach().clone(), value.grad.clone(), offsets.grad.clone(), logits.grad.clone()]
    value.grad = offsets.grad = logits.grad = None

    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    if ref_dim == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + offsets / P * (ref[:, :, None, :, None, 2::2] + ref[:, :, None, :, None, 3::2]) * 0.5
    out_u = MSDeformAttnFunction.apply(value, shapes, lsi, loc.contiguous(), aw.contiguous(), 64)
    out_u.backward(go)
    want = [out_u.detach(), value.grad, offsets.grad, logits.grad]
    for name, a, b in zip(("out", "grad_value", "grad_offsets", "grad_logits"), got, want):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), (name, ((a - b).abs().max() / b.abs().max()).item())
    # the C-ABI refuses what it does not cover
    from monosowa_amd import _lib
    assert _lib.load().msda_fused_forward_f32(1, 1, 1, 1, 1, 1, 3, 1, 1, 1, 1, 32, 4, 1, 4, 1, 1, None) == -3


@pytest.mark.parametrize("B,M,ref_dim", [(3, 5, 2), (1, 3, 2), (2, 8, 6), (2, 8, -2)])
def test_fused_self_attention_shape_odd_planes_and_6d_reference_points(B, M, ref_dim):
    """Lq == S (the window / row-tile kernels' shape) with (batch x head) counts that are not multiples of 8 -- the persistent
    workgroups walk a padded item list -- and with 6-d reference points, which the window kernels do not evaluate: the
    library must fall back to the record kernels, and msda_fused_save_supported() must say so.  Against the unfused
    operator fed with the PyTorch prologue."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedFunction
    MSDA = _msda()
    torch.manual_seed(B * 100 + M)
    levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
    D, L, P = 32, 4, 4
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
    S = Lq = int(shapes.prod(1).sum())
    centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                                    indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in levels])
    value = torch.randn(B, S, M, D, device="cuda", requires_grad=True)
    offsets = (torch.randn(B, Lq, M, L, P, 2, device="cuda") * 2.5).requires_grad_(True)
    logits = torch.randn(B, Lq, M, L * P, device="cuda", requires_grad=True)
    ref = centres[None, :, None, :].expand(B, Lq, L, 2)
    if ref_dim == -2:
        # Lq == S but the queries are NOT at their tokens' pixels: the windows and the near-point scan are then useless (almost
        # every tap is fetched from global memory, almost every point is "far" and goes through the atomics) -- never wrong
        ref_dim = 2
        ref = torch.rand(B, Lq, L, 2, device="cuda")
    if ref_dim == 6:
        ref = torch.cat([ref, torch.rand(B, Lq, L, 4, device="cuda") * 0.2], -1)
    ref = ref.contiguous()
    assert MSDA.fused_save_supported(value, shapes, lsi, Lq, ref_dim) == (ref_dim == 2 and M * L * 8 <= 1024)
    go = torch.randn(B, Lq, M * D, device="cuda")
    out_f = MSDeformAttnFusedFunction.apply(value, shapes, lsi, offsets, logits, ref)
    out_f.backward(go)
    got = [out_f.detach().clone(), value.grad.clone(), offsets.grad.clone(), logits.grad.clone()]
    value.grad = offsets.grad = logits.grad = None
    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    if ref_dim == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + offsets / P * (ref[:, :, None, :, None, 2::2] + ref[:, :, None, :, None, 3::2]) * 0.5
    out_u = MSDeformAttnFunction.apply(value, shapes, lsi, loc.contiguous(), aw.contiguous(), 64)
    out_u.backward(go)
    want = [out_u.detach(), value.grad, offsets.grad, logits.grad]
    for name, a, b in zip(("out", "grad_value", "grad_offsets", "grad_logits"), got, want):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), (name, ((a - b).abs().max() / b.abs().max()).item())


@pytest.mark.parametrize("kind", ["self", "cross"])
@pytest.mark.parametrize("masked,strided", [(True, False), (False, True), (True, True)])
def test_fused_operator_on_a_value_view_equals_masked_fill_plus_the_unfused_operator(kind, masked, strided):
    """ABI v7 (msda_fused_*_view_f32): `value` as one 256-column block of a [B, S, 768] projection (token stride) and / or a
    padding mask, on the self-attention shape (window gather + row-tile scatter, saved prologue) and the cross-attention
    shape (record gather + tile-owner scatter) -- against value.masked_fill(...) (ms_deform_attn.py:139-140) + the PyTorch
    prologue + the unfused operator.  grad_value comes back dense with exactly-zero rows for padded tokens."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedMergedFunction
    MSDA = _msda()
    torch.manual_seed(31 + masked + 2 * strided)
    levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
    B, M, D, L, P = 2, 8, 32, 4, 4
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
    S = int(shapes.prod(1).sum())
    if kind == "self":
        Lq = S
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                                    indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in levels])
        ref = ref[None, :, None, :].expand(B, Lq, L, 2).contiguous()
        off_scale = 3.0                                       # some far / out-of-window points as well
    else:
        Lq = 333
        ref = torch.cat([torch.rand(B, Lq, L, 2, device="cuda"), torch.rand(B, Lq, L, 4, device="cuda") * 0.3], -1)
        off_scale = 3.0
    wide = torch.randn(B, S, 3 * M * D, device="cuda")
    base = (wide[:, :, M * D:2 * M * D] if strided else wide[:, :, M * D:2 * M * D].contiguous()).view(B, S, M, D)
    value = base.detach().requires_grad_(True) if not strided else None
    if strided:
        wide.requires_grad_(True)
        value_in = wide[:, :, M * D:2 * M * D].view(B, S, M, D)
        assert not value_in.is_contiguous()
    else:
        value_in = value
    mask = (torch.rand(B, S, device="cuda") < 0.25) if masked else None
    proj = torch.cat([torch.randn(B, Lq, M * 32, device="cuda") * off_scale, torch.randn(B, Lq, M * 16, device="cuda")], -1).requires_grad_(True)
    go = torch.randn(B, Lq, M * D, device="cuda")
    out = MSDeformAttnFusedMergedFunction.apply(value_in, shapes, lsi, proj, ref, mask)
    out.backward(go)
    gv = (wide.grad[:, :, M * D:2 * M * D] if strided else value.grad.view(B, S, M * D)).clone()
    if strided:
        assert wide.grad[:, :, :M * D].abs().max() == 0 and wide.grad[:, :, 2 * M * D:].abs().max() == 0
    got = [out.detach().clone(), gv, proj.grad.clone()]
    proj.grad = None

    v2 = base.detach().clone().contiguous().requires_grad_(True)
    vm = v2.masked_fill(mask[..., None, None], 0.0) if masked else v2
    offsets, logits = proj[:, :, :M * 32].view(B, Lq, M, L, P, 2), proj[:, :, M * 32:].reshape(B, Lq, M, L * P)
    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    if ref.shape[-1] == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + offsets / P * (ref[:, :, None, :, None, 2::2] + ref[:, :, None, :, None, 3::2]) * 0.5
    out_u = MSDeformAttnFunction.apply(vm, shapes, lsi, loc.contiguous(), aw.contiguous(), 64)
    out_u.backward(go)
    want = [out_u.detach(), v2.grad.view(B, S, M * D), proj.grad]
    for name, a, b in zip(("out", "grad_value", "grad_proj"), got, want):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), (kind, name, ((a - b).abs().max() / b.abs().max()).item())
    if masked:
        assert got[1][mask].abs().max() == 0                  # padded tokens: exactly zero gradient rows


def test_fused_operator_hands_a_gradient_to_2d_reference_points():
    """The decoder's first layer: 2-d reference points derived from the learned query embedding carry a gradient
    (ms_deform_attn.py:149-152: location = ref + offset / (W, H)).  The fused merged operator derives it from d offsets;
    against autograd through the PyTorch prologue + the unfused operator."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedMergedFunction
    MSDA = _msda()
    torch.manual_seed(41)
    levels = [(24, 40), (12, 20), (6, 10), (3, 5)]
    B, M, D, L, P, Lq = 2, 8, 32, 4, 4, 275
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, device="cuda", requires_grad=True)
    pts = torch.rand(B, Lq, 2, device="cuda", requires_grad=True)
    ratios = torch.rand(B, L, 2, device="cuda") * 0.2 + 0.8
    proj = torch.cat([torch.randn(B, Lq, M * 32, device="cuda") * 2, torch.randn(B, Lq, M * 16, device="cuda")], -1).requires_grad_(True)
    go = torch.randn(B, Lq, M * D, device="cuda")
    ref = pts[:, :, None] * ratios[:, None]                               # depthaware_transformer.py:590-596
    out = MSDeformAttnFusedMergedFunction.apply(value, shapes, lsi, proj, ref.contiguous())
    out.backward(go)
    got = [out.detach().clone(), value.grad.clone(), proj.grad.clone(), pts.grad.clone()]
    value.grad = proj.grad = pts.grad = None
    ref = pts[:, :, None] * ratios[:, None]
    offsets, logits = proj[:, :, :M * 32].view(B, Lq, M, L, P, 2), proj[:, :, M * 32:].reshape(B, Lq, M, L * P)
    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    out_u = MSDeformAttnFunction.apply(value, shapes, lsi, loc.contiguous(), aw.contiguous(), 64)
    out_u.backward(go)
    want = [out_u.detach(), value.grad, proj.grad, pts.grad]
    for name, a, b in zip(("out", "grad_value", "grad_proj", "grad_reference_points"), got, want):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), (name, ((a - b).abs().max() / b.abs().max()).item())


def test_self_attention_shape_beyond_the_window_kernels_addressing_range():
    """The tile-window kernels address a (batch, head) plane with 32-bit offsets: S < 2^17 tokens.  A larger pyramid at the
    self-attention shape must take the record kernels (saved prologue reported unsupported) and still equal the unfused
    operator."""
    from monosowa_amd.ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedMergedFunction
    MSDA = _msda()
    torch.manual_seed(47)
    levels = [(256, 416), (128, 208), (64, 104), (32, 52)]            # S = 141,440 >= 2^17
    B, M, D, L, P = 1, 2, 32, 4, 4
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
    S = int(shapes.prod(1).sum())
    assert S >= 1 << 17
    Lq = S
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                                indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in levels])
    ref = ref[None, :, None, :].expand(B, Lq, L, 2).contiguous()
    value = torch.randn(B, S, M, D, device="cuda", requires_grad=True)
    assert not MSDA.fused_save_supported(value, shapes, lsi, Lq, 2)
    proj = torch.cat([torch.randn(B, Lq, M * 32, device="cuda") * 3, torch.randn(B, Lq, M * 16, device="cuda")], -1).requires_grad_(True)
    go = torch.randn(B, Lq, M * D, device="cuda")
    out = MSDeformAttnFusedMergedFunction.apply(value, shapes, lsi, proj, ref)
    out.backward(go)
    got = [out.detach().clone(), value.grad.clone(), proj.grad.clone()]
    value.grad = proj.grad = None
    offsets, logits = proj[:, :, :M * 32].view(B, Lq, M, L, P, 2), proj[:, :, M * 32:].reshape(B, Lq, M, L * P)
    aw = torch.softmax(logits, -1).view(B, Lq, M, L, P)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    out_u = MSDeformAttnFunction.apply(value, shapes, lsi, loc.contiguous(), aw.contiguous(), 64)
    out_u.backward(go)
    for name, a, b in zip(("out", "grad_value", "grad_proj"), got, [out_u.detach(), value.grad, proj.grad]):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), (name, ((a - b).abs().max() / b.abs().max()).item())


def test_train_val_cli_runs_an_epoch_and_writes_kitti_results(tmp_path):
    """tools/train_val.py (the reference's CLI): one tiny epoch on synthetic data through Trainer -> checkpoint ->
    Tester.inference -> KITTI result files, then `-e` evaluation-only from the saved checkpoint."""
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "monodetr.yaml")))
    cfg["dataset"].update(batch_size=2, resolution=[320, 96], num_samples=4)
    cfg["trainer"].update(max_epoch=1, save_path="out/")        # the reference prefixes './' (trainer_helper.py:40)
    cfg["tester"].update(threshold=0.0)
    path = tmp_path / "tiny.yaml"
    yaml.safe_dump(cfg, open(path, "w"))
    env = dict(os.environ, PYTHONPATH=root)
    for extra in ([], ["-e"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "train_val.py"), "--config", str(path), "--workers", "0"] + extra,
                           cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / "out" / "monodetr"
    assert (out / "checkpoint_epoch_1.pth").exists()
    ckpt = torch.load(out / "checkpoint_epoch_1.pth", map_location="cpu", weights_only=False)
    assert set(ckpt) == {"epoch", "model_state", "optimizer_state", "best_result", "best_epoch"} and ckpt["epoch"] == 1
    files = sorted((out / "outputs" / "data").glob("*.txt"))
    assert len(files) == 4
    line = files[0].read_text().splitlines()[0].split(" ")
    assert line[0] in ("Pedestrian", "Car", "Cyclist") and len(line) == 16


def test_pointwise_bias_residual_relu_kernel():
    """mono_bias_act_f32 / mono_relu_grad_f32 against the PyTorch formulation, forward and gradients."""
    from monosowa_amd.pointwise import bias_act
    torch.manual_seed(0)
    for res in (False, True):
        for relu in (True, False):
            y0 = torch.randn(3, 64, 17, 23, device="cuda").contiguous(memory_format=torch.channels_last)
            b = torch.randn(64, device="cuda")
            r = torch.randn_like(y0).contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
            src = y0.clone().requires_grad_(True)
            out = bias_act(src * 1.0, b, r, relu)
            go = torch.randn_like(out)
            out.backward(go)
            ref_src = y0.clone().requires_grad_(True)
            ref_r = r.detach().clone().requires_grad_(True) if res else None
            ref = ref_src + b.view(1, -1, 1, 1) + (ref_r if res else 0)
            ref = torch.relu(ref) if relu else ref
            ref.backward(go)
            assert torch.equal(out, ref)
            assert torch.equal(src.grad, ref_src.grad)
            if res:
                assert torch.equal(r.grad, ref_r.grad)


def test_fused_dropout_add_layernorm():
    """mono_dropout_add_layernorm_*: exact LayerNorm(x + z) parity without dropout (forward + all gradients); with
    dropout the kept fraction matches p, kept elements are scaled by 1/(1-p), and the backward reuses the same mask."""
    from monosowa_amd.pointwise import dropout_add_layernorm
    torch.manual_seed(0)
    norm = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = torch.randn(7, 1021, 256, device="cuda", requires_grad=True)
    z = torch.randn(7, 1021, 256, device="cuda", requires_grad=True)
    go = torch.randn(7, 1021, 256, device="cuda")
    drop = torch.nn.Dropout(0.1).cuda().eval()
    y = dropout_add_layernorm(x, z, norm, drop)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), z.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    x.grad = z.grad = None
    norm.zero_grad()
    ref = norm(x + z)
    ref.backward(go)
    for a, b, n in zip(got, [ref.detach(), x.grad, z.grad, norm.weight.grad, norm.bias.grad], ("y", "gx", "gz", "gw", "gb")):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), n
    # training mode: recover the mask from gz / gx (gz = gx * keep / (1 - p))
    drop.train()
    x.grad = z.grad = None
    y = dropout_add_layernorm(x, z, norm, drop)
    y.backward(go)
    ratio = z.grad / x.grad
    kept = ratio.abs() > 0.5
    assert abs(kept.float().mean().item() - 0.9) < 0.005
    assert torch.allclose(ratio[kept], torch.full_like(ratio[kept], 1 / 0.9), rtol=1e-4)
    mask = kept.float() / 0.9
    ref = norm(x.detach() + z.detach() * mask)
    assert (y - ref).abs().max() <= 2e-5 * ref.abs().max()
    y2 = dropout_add_layernorm(x, z, norm, drop)          # a new call draws a new mask
    assert not torch.equal(y2, y)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(3, 256, 24, 80), (2, 256, 7, 9), (1, 256, 1, 1)])
def test_groupnorm_nhwc_matches_torch(shape, relu):
    """mono_groupnorm_nhwc_*: GroupNorm(32, 256) (+ ReLU) on channels-last tensors equals F.group_norm (forward,
    grad_input, grad_weight, grad_bias), including a large-mean input (f64 statistics) and ragged pixel counts."""
    from monosowa_amd.pointwise import group_norm
    torch.manual_seed(1)
    gn = torch.nn.GroupNorm(32, 256).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(shape, device="cuda") * 2 + 5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
    y = group_norm(x, gn, relu=relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(go)
    got = [y.detach().clone(), x.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()]
    x.grad = None
    gn.zero_grad()
    xd = x.detach().double().contiguous().requires_grad_(True)
    gnd = torch.nn.GroupNorm(32, 256).cuda().double()
    gnd.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    ref = gnd(xd)
    ref = torch.relu(ref) if relu else ref
    ref.backward(go.double())
    for a, b, n in zip(got, [ref.detach(), xd.grad, gnd.weight.grad, gnd.bias.grad], ("y", "gx", "gw", "gb")):
        assert (a.double() - b).abs().max() <= 3e-5 * max(b.abs().max().item(), 1.0), n


@pytest.mark.parametrize("rows,C", [(8800, 256), (163200, 128), (30720, 512), (777, 1024), (5, 24), (1, 4), (30720, 81), (8800, 3), (131, 1023)])
def test_colsum_matches_torch(rows, C):
    """mono_colsum_f32 (bias gradients) equals a float64 column sum."""
    from monosowa_amd.pointwise import colsum
    torch.manual_seed(rows)
    g = torch.randn(rows, C, device="cuda")
    ref = g.double().sum(0)
    got = colsum(g)
    assert got.shape == (C,)
    assert (got.double() - ref).abs().max() <= 2e-6 * g.abs().double().sum(0).max()


@pytest.mark.parametrize("B,S,C,bounds", [(16, 10200, 384, [(0, 7680), (7680, 9600), (9600, 10080), (10080, 10200)]),
                                          (4, 777, 256, [(0, 1), (1, 400), (500, 777)]),
                                          (2, 300, 128, [(0, 300)]),
                                          (3, 1000, 512, [(10, 20), (0, 1000), (999, 1000), (5, 6), (100, 900), (0, 64), (64, 128), (128, 129)])])
def test_colsum_levels_one_launch_equals_float64_and_the_per_level_launches(B, S, C, bounds):
    """pointwise.colsum_levels (mono_colsum_levels_f32: the encoder's per-pyramid-level sums of d proj, depthaware_transformer.py:232-240):
    one launch pair over all levels against float64, and against the launch-pair-per-level form it replaces (overlapping and
    gapped ranges included: the ranges are independent)."""
    from monosowa_amd import pointwise as pw
    torch.manual_seed(B * S + C)
    g = torch.randn(B, S, C, device="cuda")
    ref = torch.stack([g[:, a:b].double().sum((0, 1)) for a, b in bounds])
    got = pw.colsum_levels(g, bounds)
    assert got.shape == (len(bounds), C)
    scale = max(g.abs().double().sum((0, 1)).max().item(), 1.0)
    assert (got.double() - ref).abs().max() <= 2e-6 * scale
    try:
        pw.COLSUM_LEVELS_ONE_LAUNCH = 0
        per_level = pw.colsum_levels(g, bounds)
    finally:
        pw.COLSUM_LEVELS_ONE_LAUNCH = 1
    assert (per_level.double() - ref).abs().max() <= 2e-6 * scale
    assert torch.equal(pw.colsum_levels(g, bounds), got)               # fixed summation order
    lv, total = pw.colsum_levels(g, bounds, with_total=True)           # the sum over the levels' sums from the same launch pair
    assert torch.equal(lv, got) and total.shape == (C,)
    assert (total.double() - ref.sum(0)).abs().max() <= 2e-6 * max(ref.abs().sum(0).max().item(), scale)


def test_sum_slices_and_channel_bias():
    """pointwise.sum_slices (the slices of a split-K weight gradient) and conv_channel_bias (bias-free convolution + bias node
    whose gradient is a row sum of the channels-last gradient, odd channel counts included) against plain PyTorch."""
    from monosowa_amd.pointwise import conv_channel_bias, sum_slices
    torch.manual_seed(8)
    t = torch.randn(64, 256, 384, device="cuda")
    assert (sum_slices(t) - t.double().sum(0).float()).abs().max() <= 2e-5
    t = torch.randn(3, 5, 8, device="cuda")
    assert (sum_slices(t) - t.sum(0)).abs().max() <= 1e-6
    conv = torch.nn.Conv2d(32, 81, 1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(4, 32, 24, 80, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(4, 81, 24, 80, device="cuda").contiguous(memory_format=torch.channels_last)
    res = []
    for f in (lambda: conv_channel_bias(conv, x), lambda: conv(x)):
        x.grad = conv.weight.grad = conv.bias.grad = None
        y = f()
        (y * go).sum().backward()
        res.append([y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()])
    for a, b in zip(*res):
        assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-3)


# ---- fp32 attention (libmonosowa_attn.so) ------------------------------------------------------------------
def _heads(L, B, H):
    return torch.randn(L, B, H * 32, device="cuda").view(L, B, H, 32).permute(1, 2, 0, 3)


@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 8, 100, 200), (1, 3, 33, 64), (2, 8, 550, 1920), (1, 8, 1920, 1920), (3, 2, 1, 5)])
def test_attention_matches_fp64_softmax_attention(B, H, Lq, Lk):
    """mono_attn_*: O, dQ, dK, dV against softmax(QK^T/sqrt(32))V evaluated in float64 (ragged Lq / Lk included)."""
    import math
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(Lq + Lk)
    q, k, v = _heads(Lq, B, H), _heads(Lk, B, H), _heads(Lk, B, H)
    go = _heads(Lq, B, H)
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    o = FA.attention(qg, kg, vg)
    o.backward(go)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax((qd @ kd.transpose(-1, -2)) / math.sqrt(32), -1) @ vd
    ref.backward(go.double())
    for name, a, b in (("o", o, ref), ("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        assert (a.double() - b).abs().max() <= 1e-5 * max(b.abs().max().item(), 1e-3), name


@pytest.mark.parametrize("masked", [False, True], ids=["nomask", "keypad"])
def test_attention_dropout_uses_one_mask_forward_and_backward(masked):
    """The dropped probabilities are recovered with one-hot V; the kernel gradients must equal float64 autograd through
    exactly that mask, the kept fraction must match p, and a different seed must give a different mask.  With a key padding
    mask (the depth-aware decoder's cross attention over the padded image memory) the padded keys carry no probability and the
    dropout mask is the one the unmasked kernels draw."""
    import math
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(1)
    B, H, Lq, Lk, p, seed = 2, 3, 70, 64, 0.3, 99
    q, k, v, go = _heads(Lq, B, H), _heads(Lk, B, H), _heads(Lk, B, H), _heads(Lq, B, H)
    kpm = None
    if masked:
        kpm = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
        kpm[0, 50:] = True
        kpm[1, ::7] = True
    scale = 1 / math.sqrt(32)
    eye = torch.eye(64, device="cuda")
    recover = lambda sd: torch.cat([FA.forward(q, k, eye[:, i * 32:(i + 1) * 32].expand(B, H, 64, 32).contiguous(), scale, p, sd,
                                               key_padding_mask=kpm)[0] for i in range(2)], -1)
    pd = recover(seed)
    live = torch.ones(B, 1, 1, Lk, device="cuda", dtype=torch.bool) if kpm is None else ~kpm[:, None, None, :]
    assert (pd[~live.expand_as(pd)] == 0).all(), "padded keys must carry no probability"
    mask = (pd != 0).double()
    kept = mask[live.expand_as(mask)].mean().item()
    assert abs(kept - (1 - p)) < 0.02
    assert not torch.equal(recover(seed + 1) != 0, pd != 0)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    keep_scale = 65536.0 / (65536.0 - round(p * 65536))
    logits = (qd @ kd.transpose(-1, -2)) * scale
    if kpm is not None:
        logits = logits.masked_fill(~live, float("-inf"))
    pr = torch.softmax(logits, -1) * mask * keep_scale
    assert (pd.double() - pr).abs().max() < 1e-6
    (pr @ vd).backward(go.double())
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    FA.attention(qg, kg, vg, dropout_p=p, seed=seed, key_padding_mask=kpm).backward(go)
    for name, a, b in (("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        assert (a.double() - b).abs().max() <= 1e-5 * b.abs().max(), name
    if kpm is not None:
        assert (kg.grad[kpm[:, None, :, None].expand_as(kg.grad)] == 0).all() and (vg.grad[kpm[:, None, :, None].expand_as(vg.grad)] == 0).all()


@pytest.mark.parametrize("masked", [False, True], ids=["nomask", "keypad"])
@pytest.mark.parametrize("B,H,Lq,Lk", [(2, 8, 550, 1920), (1, 3, 70, 130), (2, 2, 300, 64), (1, 8, 1920, 1920), (3, 2, 1, 5)])
def test_attention_backward_from_saved_keep_bits_is_the_rehashed_backward(B, H, Lq, Lk, masked):
    """The training path hands the forward's dropout mask to the backward as one bit per score (mono_attn_*_keep_f32) instead of
    re-hashing it there: the three gradients must be BIT-identical to the re-hashing backward (same mask, same arithmetic), for
    ragged query / key counts, with and without a key padding mask."""
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(Lq + Lk)
    q, k, v, go = _heads(Lq, B, H), _heads(Lk, B, H), _heads(Lk, B, H), _heads(Lq, B, H)
    kpm = None
    if masked:
        kpm = torch.rand(B, Lk, device="cuda") < 0.2
        kpm[:, 0] = False
    p, seed, scale = 0.1, 4242, 1 / 32 ** 0.5
    bits = FA.keep_bits_like(q, k, p)
    bits.fill_(-1)
    o1, lse1 = FA.forward(q, k, v, scale, p, seed, key_padding_mask=kpm, keep_bits=bits)
    o2, lse2 = FA.forward(q, k, v, scale, p, seed, key_padding_mask=kpm)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    go = torch.empty_like(o1).copy_(go)                        # (the backward wants dout laid out like o)
    with_bits = FA.backward(q, k, v, o1, lse1, go, scale, p, seed, key_padding_mask=kpm, keep_bits=bits)
    rehashed = FA.backward(q, k, v, o1, lse1, go, scale, p, seed, key_padding_mask=kpm)
    for name, a, b in zip(("dq", "dk", "dv"), with_bits, rehashed):
        assert torch.equal(a, b), "%s: max difference %.3e" % (name, (a - b).abs().max().item())


@pytest.mark.parametrize("same_qk", [True, False, None])
def test_mha_forward_equals_nn_multiheadattention(same_qk):
    """mha_forward (packed projections + HIP core) against nn.MultiheadAttention in eval mode: output and all
    parameter / input gradients, for the depth-encoder (q is k) and decoder (k is v) call patterns."""
    from monosowa_amd.flash_attn import mha_forward, mha_supported
    torch.manual_seed(3)
    mha = torch.nn.MultiheadAttention(256, 8, dropout=0.1).cuda().eval()
    x = torch.randn(96, 2, 256, device="cuda", requires_grad=True)
    y = torch.randn(40, 2, 256, device="cuda", requires_grad=True)
    if same_qk is None:            # decoder self-attention: three different tensors, folded groups (non-contiguous views)
        z = torch.randn(2, 96, 256, device="cuda", requires_grad=True)
        args = (z.transpose(0, 1), (x * 1.5), (x.detach() * 0.5).clone().requires_grad_(True))
        args = (args[0], args[1].detach().requires_grad_(True), args[2])
    else:
        args = (x, x, (x.detach() * 0.5).clone().requires_grad_(True)) if same_qk else (y, x, x)
    assert mha_supported(mha, *args)
    go = torch.randn(args[0].shape, device="cuda")

    def run(fn):
        for t in set(args):
            (t if t.is_leaf else t._base).grad = None
        mha.zero_grad()
        out = fn()
        out.backward(go)
        leaves = [t if t.is_leaf else t._base for t in args]
        return [out.detach().clone()] + [t.grad.clone() for t in leaves if t.grad is not None] + [p_.grad.clone() for p_ in mha.parameters()]
    ours = run(lambda: mha_forward(mha, *args))
    ref = run(lambda: mha(*args, need_weights=False)[0])
    assert len(ours) == len(ref)
    for a, b in zip(ours, ref):
        assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-3)


@pytest.mark.parametrize("pattern", ["depth_encoder", "decoder_cross"])
@pytest.mark.parametrize("train", [False, True])
def test_mha_forward_with_padded_keys_equals_nn_multiheadattention(pattern, train):
    """The key padding mask inside the HIP attention kernels (mono_attn_*_masked_f32): mha_forward(..., key_padding_mask) against
    nn.MultiheadAttention(..., key_padding_mask) -- output, input and parameter gradients -- with a different number of padded
    keys per batch element, a padded run that ends inside a 64-key tile, a ragged last tile, and one element without padding.
    Reference call sites: depthaware_transformer.py:456-459 (decoder depth cross-attention), depth_predictor/transformer.py:57-60.
    train = True: dropout 0 in train mode (the module's dropout path with p = 0; a random mask cannot be compared across
    implementations -- test_attention_dropout_uses_one_mask_forward_and_backward covers it)."""
    from monosowa_amd.flash_attn import mha_forward, mha_supported
    torch.manual_seed(5)
    mha = torch.nn.MultiheadAttention(256, 8, dropout=0.0).cuda().train(train)
    B, Lk, Lq = 3, 150, 70
    x = torch.randn(Lk, B, 256, device="cuda", requires_grad=True)
    y = torch.randn(Lq, B, 256, device="cuda", requires_grad=True)
    mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    mask[0, 100:] = True            # a padded tail that starts inside the second tile
    mask[1, 5:40] = True            # ... and a padded run in the middle
    args = (x, x, (x.detach() * 0.5).clone().requires_grad_(True)) if pattern == "depth_encoder" else (y, x, x)
    assert mha_supported(mha, *args)
    go = torch.randn(args[0].shape, device="cuda")

    def run(fn):
        for t in set(args):
            t.grad = None
        mha.zero_grad()
        out = fn()
        out.backward(go)
        return [out.detach().clone()] + [t.grad.clone() for t in args if t.grad is not None] + [p_.grad.clone() for p_ in mha.parameters()]
    ours = run(lambda: mha_forward(mha, *args, key_padding_mask=mask))
    ref = run(lambda: mha(*args, key_padding_mask=mask, need_weights=False)[0])
    assert len(ours) == len(ref)
    for a, b in zip(ours, ref):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-3)


def test_masked_attention_core_matches_fp64_and_zeroes_the_padded_keys_gradients():
    """attention(q, k, v, key_padding_mask) against an fp64 masked softmax attention; dk / dv of padded keys are exactly zero."""
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(6)
    B, H, Lq, Lk = 2, 8, 200, 333
    q, k, v = (torch.randn(B, H, L, 32, device="cuda", requires_grad=True) for L in (Lq, Lk, Lk))
    mask = torch.rand(B, Lk, device="cuda") < 0.3
    mask[:, 0] = False
    go = torch.randn(B, H, Lq, 32, device="cuda")
    out = FA.attention(q, k, v, key_padding_mask=mask)
    out.backward(go)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(-1, -2)) / 32 ** 0.5
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = torch.softmax(s, -1) @ vd
    ref.backward(go.double())
    assert (out.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    for name, a, b in (("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        assert (a.double() - b).abs().max() <= 1e-5 * b.abs().max(), name
    dead = mask[:, None, :, None].expand_as(k.grad)
    assert (k.grad[dead] == 0).all() and (v.grad[dead] == 0).all()


def test_masked_attention_with_whole_key_tiles_padded_in_front_of_valid_keys_matches_fp64():
    """A 64-key tile whose keys are ALL padded in FRONT of valid keys (and one in the middle): the running maximum is still -inf when
    that tile is seen, and (-inf) - (-inf) must not poison the row (torch gives finite output).  Forward and backward
    against an fp64 masked softmax; the model's own right / bottom padding never produces this, general key_padding_mask callers can."""
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(9)
    B, H, Lq, Lk = 3, 8, 150, 300
    q, k, v = (torch.randn(B, H, L, 32, device="cuda", requires_grad=True) for L in (Lq, Lk, Lk))
    mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    mask[0, :64] = True                       # the leading tile
    mask[1, :130] = True                      # two leading tiles and a bit
    mask[1, 192:256] = True                   # and a whole tile in the middle
    mask[2, 64:128] = True                    # a middle tile only
    go = torch.randn(B, H, Lq, 32, device="cuda")
    out = FA.attention(q, k, v, key_padding_mask=mask)
    out.backward(go)
    assert torch.isfinite(out).all()
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(-1, -2)) / 32 ** 0.5
    ref = torch.softmax(s.masked_fill(mask[:, None, None, :], float("-inf")), -1) @ vd
    ref.backward(go.double())
    assert (out.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    for name, a, b in (("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        assert torch.isfinite(a).all(), name
        assert (a.double() - b).abs().max() <= 1e-5 * b.abs().max(), name
    # with dropout the same rows stay finite (the keep-bit and the re-hashing backward)
    o2 = FA.attention(q.detach(), k.detach(), v.detach(), dropout_p=0.1, seed=5, key_padding_mask=mask)
    assert torch.isfinite(o2).all()


def test_masked_attention_with_every_key_of_one_image_padded_is_nan_there_like_torch_and_exact_elsewhere():
    """A batch element whose keys are ALL padded has no softmax (torch: NaN rows, F.multi_head_attention_forward's masked_fill(-inf) ->
    softmax); the kernels say the same instead of inventing a value, and the other batch elements do not notice."""
    from monosowa_amd import flash_attn as FA
    torch.manual_seed(8)
    B, H, Lq, Lk = 3, 8, 70, 130
    q, k, v = (torch.randn(B, H, L, 32, device="cuda") for L in (Lq, Lk, Lk))
    mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    mask[1] = True
    mask[2, 100:] = True
    out = FA.attention(q, k, v, key_padding_mask=mask)
    s = (q.double() @ k.double().transpose(-1, -2)) / 32 ** 0.5
    ref = torch.softmax(s.masked_fill(mask[:, None, None, :], float("-inf")), -1) @ v.double()
    assert torch.isnan(ref[1]).all() and torch.isnan(out[1]).all()
    for b in (0, 2):
        assert (out[b].double() - ref[b]).abs().max() <= 1e-5 * ref[b].abs().max()


def test_fused_adamw_matches_the_foreach_formulation():
    """mono_adamw_step_f32 (one launch for all parameters) against the foreach evaluation of the same update
    (itself bit-identical to the reference on the CPU, tests/test_helpers.py), over several steps, with odd sizes,
    a weight-decay group, and unaligned gradient views."""
    from monosowa_amd.helpers.optimizer_helper import AdamW
    torch.manual_seed(0)
    shapes = [(7,), (256, 256, 3, 3), (33, 5), (1,), (70001,), (128,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pa[1] = torch.nn.Parameter(pa[1].detach().contiguous(memory_format=torch.channels_last))      # like a conv weight
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    mk = lambda ps: AdamW([{"params": ps[:2], "weight_decay": 0}, {"params": ps[2:], "weight_decay": 1e-4}], lr=2e-4)
    fused, plain = mk(pa), mk(pb)
    plain._fused_step = lambda *a, **k: False
    pool = torch.randn(sum(p.numel() for p in pa) + 1, device="cuda")
    for it in range(4):
        off = 1                                              # 4-byte aligned only: exercises the scalar path
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device="cuda") * (10.0 ** (it - 2))
            pool[off:off + a.numel()] = g.flatten()
            g = g.contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g
            a.grad = pool[off:off + a.numel()].view(a.shape) if (it % 2 and g.dim() != 4) else g.clone()
            b.grad = g.clone()
            off += a.numel()
        fused.step()
        plain.step()
        for a, b in zip(pa, pb):
            assert (a - b).abs().max() <= 1e-6 * b.abs().max()
            assert (fused.state[a]["exp_avg_sq"] - plain.state[b]["exp_avg_sq"]).abs().max() <= 1e-6 * plain.state[b]["exp_avg_sq"].abs().max()
    assert getattr(fused, "_fused_plans", None), "the fused path did not run"


def test_resnet_stage_with_forked_relu_backward_matches_plain_autograd():
    """Two bottleneck blocks through the product path (folded frozen BN, in-place bias+residual+ReLU, the block
    output handed on as a pair whose gradients meet in mono_relu_grad2_f32) against the same blocks written with
    plain PyTorch ops: output, input gradient and every convolution weight gradient."""
    import torch.nn.functional as F
    from monosowa_amd.monodetr.backbone import Bottleneck, FrozenBatchNorm2d
    torch.manual_seed(5)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=2, bias=False), FrozenBatchNorm2d(128))
    blocks = torch.nn.Sequential(Bottleneck(64, 32, stride=2, downsample=ds), Bottleneck(128, 32), Bottleneck(128, 32)).cuda()
    for m in blocks.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3); m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
    blocks = blocks.to(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 24, 40, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(2, 128, 12, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    params = [p for p in blocks.parameters() if p.requires_grad]

    def plain(x):
        def cb(x, conv, bn, res=None, relu=True):
            y = bn(conv(x))
            y = y if res is None else y + res
            return F.relu(y) if relu else y
        for b in blocks:
            out = cb(cb(x, b.conv1, b.bn1), b.conv2, b.bn2)
            idt = x if b.downsample is None else cb(x, b.downsample[0], b.downsample[1], relu=False)
            x = cb(out, b.conv3, b.bn3, idt)
        return x

    def run(fn):
        x.grad = None
        for p in params:
            p.grad = None
        y = fn(x)
        (y * go).sum().backward()
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params]
    ours = run(lambda t: blocks(t)[0])
    ref = run(plain)
    assert len(ours) == len(ref) and len(ours) > 10
    for a, b in zip(ours, ref):
        assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-3)

    # a stage output with THREE consumers (the next stage's first convolution, its identity branch, the pyramid projection):
    # the last block hands out three handles, their gradients meet in mono_relu_grad_mask3_f32
    tail = Bottleneck(128, 32).cuda().to(memory_format=torch.channels_last)
    side = torch.nn.Conv2d(128, 16, 1).cuda().to(memory_format=torch.channels_last)
    params3 = params + [p for p in tail.parameters() if p.requires_grad] + list(side.parameters())
    go2 = torch.randn(2, 16, 12, 20, device="cuda").contiguous(memory_format=torch.channels_last)

    def run3(three):
        x.grad = None
        for p in params3:
            p.grad = None
        blocks[-1].n_out = 3 if three else 2
        if three:
            h = blocks(x)
            assert len(h) == 3
            y, z = tail(h[:2])[0], side(h[2])
        else:
            h = plain(x)
            y, z = tail(h)[0], side(h)
        ((y * go).sum() + (z * go2).sum()).backward()
        blocks[-1].n_out = 2
        return [y.detach().clone(), z.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params3]
    ours3, ref3 = run3(True), run3(False)
    for a, b in zip(ours3, ref3):
        assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-3)


def test_fused_relu_dropout():
    """mono_relu_dropout_*: kept fraction, 1/(1-p) scaling of the kept positives, zeros for negatives, and a backward
    that uses exactly the forward's mask."""
    from monosowa_amd.pointwise import relu_dropout
    torch.manual_seed(0)
    drop = torch.nn.Dropout(0.1).cuda().train()
    h = torch.randn(16, 1024, 256, device="cuda", requires_grad=True)
    go = torch.randn_like(h)
    y = relu_dropout(h, drop)
    assert "ReluDropout" in type(y.grad_fn).__name__
    y.backward(go)
    pos = h.detach() > 0
    kept = y.detach() != 0
    assert not (kept & ~pos).any()
    assert abs((kept & pos).float().sum().item() / pos.float().sum().item() - 0.9) < 0.003
    assert torch.allclose(y.detach()[kept], h.detach()[kept] / 0.9, rtol=1e-6)
    assert torch.allclose(h.grad[kept], go[kept] / 0.9, rtol=1e-6) and (h.grad[~kept] == 0).all()
    drop.eval()
    assert torch.equal(relu_dropout(h, drop), torch.relu(h))


@pytest.mark.parametrize("relu", [False, True])
def test_conv_group_norm_with_folded_bias_matches_sequential(relu):
    """conv_group_norm: bias-free convolution + GroupNorm kernels carrying the convolution bias, against
    nn.Sequential(Conv2d, GroupNorm) (+ ReLU): output, input gradient, conv weight / bias and norm weight / bias gradients."""
    from monosowa_amd.pointwise import conv_group_norm
    torch.manual_seed(2)
    conv = torch.nn.Conv2d(64, 256, 3, padding=1).cuda().to(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(32, 256).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-1, 1); gn.weight.uniform_(0.5, 1.5); gn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(3, 64, 13, 21, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(3, 256, 13, 21, device="cuda").contiguous(memory_format=torch.channels_last)
    params = list(conv.parameters()) + list(gn.parameters())

    def run(fn):
        x.grad = None
        for p_ in params:
            p_.grad = None
        y = fn()
        y.backward(go)
        return [y.detach().clone(), x.grad.clone()] + [p_.grad.clone() for p_ in params]
    ours = run(lambda: conv_group_norm(x, conv, gn, relu=relu))
    ref = run(lambda: torch.relu(gn(conv(x))) if relu else gn(conv(x)))
    for a, b, n in zip(ours, ref, ("y", "gx", "gw_conv", "gb_conv", "gw_gn", "gb_gn")):
        assert (a - b).abs().max() <= 5e-5 * max(b.abs().max().item(), 1e-3), n


def test_fused_matched_losses_equal_the_pytorch_formulation():
    """mono_matched_losses_*: every loss of the criterion and every gradient with the fused matched-pair kernels against
    the PyTorch formulation of the same criterion (itself checked against the layer-wise reference formulation on the CPU)."""
    import yaml
    from monosowa_amd.helpers.model_helper import build_model
    from monosowa_amd.monodetr import criterion as C
    from monosowa_amd.synthetic import make_batch, prepare_targets
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "monodetr.yaml")))
    torch.manual_seed(0)
    _, crit = build_model(dict(cfg["model"], device="cuda"))
    crit = crit.cuda().train()
    B, Q = 4, 550
    _, _, targets, _ = make_batch(B, "cuda", seed=5, resolution=(320, 96))
    tl = prepare_targets(targets, B)
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g)

    def outputs():
        layer = lambda: {"pred_logits": mk(B, Q, 3).requires_grad_(True), "pred_boxes": (mk(B, Q, 6) * 0.5).sigmoid().requires_grad_(True),
                         "pred_3d_dim": (mk(B, Q, 3) * 0.2 + 2).requires_grad_(True), "pred_depth": mk(B, Q, 2).requires_grad_(True),
                         "pred_angle": mk(B, Q, 24).requires_grad_(True)}
        o = layer()
        o["aux_outputs"] = [layer(), layer()]
        o["pred_depth_map_logits"] = mk(B, 81, 6, 20).requires_grad_(True)
        return o
    out = outputs()
    leaves = [v for d in [out] + out["aux_outputs"] for k, v in d.items() if torch.is_tensor(v)]

    def run(fused):
        C.FUSED_MATCHED = fused
        for t in leaves:
            t.grad = None
        losses = crit(out, tl)
        C.weighted_total(losses, crit.weight_dict).backward()
        return {k: v.detach().clone() for k, v in losses.items()}, [t.grad.clone() if t.grad is not None else None for t in leaves]
    try:
        l1, g1 = run(True)
        l0, g0 = run(False)
    finally:
        C.FUSED_MATCHED = True
    assert set(l1) == set(l0)
    for k in l0:
        assert abs(l1[k].item() - l0[k].item()) <= 2e-5 * max(abs(l0[k].item()), 1.0), k
    for a, b in zip(g1, g0):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max() <= 2e-5 * max(b.abs().max().item(), 1e-6)


def test_block_cost_pass_is_bitwise_on_the_gpu_too():
    from monosowa_amd.monodetr import matcher as M
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "monodetr.yaml")))["model"]
    m = M.build_matcher(cfg)
    g = torch.Generator(device="cuda").manual_seed(4)
    NL, B, Q = 3, 16, 550
    logits = torch.randn(NL, B, Q, 3, device="cuda", generator=g)
    boxes = torch.rand(NL, B, Q, 6, device="cuda", generator=g) * 0.4 + 0.1
    sizes = [int(x) for x in torch.randint(0, 12, (B,)).tolist()]
    T = sum(sizes)
    flat = {"labels": torch.randint(0, 3, (T,), device="cuda", generator=g), "boxes_3d": torch.rand(T, 6, device="cuda", generator=g) * 0.4 + 0.1}
    out = {}
    for flag in (False, True):
        M.BLOCK_COST = flag
        try:
            out[flag] = m.match_layers_end_flat(m.match_layers_begin(logits, boxes, flat, sizes, 11))
        finally:
            M.BLOCK_COST = True
    to_np = lambda x: x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)      # device solver: a CUDA tensor; host solver: numpy / pinned
    assert np.array_equal(to_np(out[True]), to_np(out[False]))


def test_encoder_block_nodes_equal_the_module_by_module_layer():
    """encoder_block: one visual-encoder layer as two autograd nodes (in-place accumulated shared gradients) against the
    same layer evaluated module by module -- eval-like (dropout 0) so both see identical arithmetic: output, input and
    positional gradients, every parameter gradient."""
    from monosowa_amd import MultiScaleDeformableAttention as MSDA
    from monosowa_amd.monodetr import depthaware_transformer as T
    torch.manual_seed(0)
    layer = T.VisualEncoderLayer(256, 256, 0.0, "relu", 4, 8, 4).cuda().train()
    levels = [(24, 80), (12, 40), (6, 20), (3, 10)]
    S = sum(h * w for h, w in levels)
    B = 16
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    starts = [0]
    for h, w in levels[:-1]:
        starts.append(starts[-1] + h * w)
    MSDA.attach_host_geometry(shapes, lsi, levels, starts)
    src = torch.randn(B, S, 256, device="cuda", requires_grad=True)
    pos = torch.randn(B, S, 256, device="cuda", requires_grad=True)
    ref = torch.rand(B, S, 4, 2, device="cuda")
    go = torch.randn(B, S, 256, device="cuda")
    params = list(layer.parameters())

    def run(flag):
        T.ENCODER_BLOCKS = flag
        for t in [src, pos] + params:
            t.grad = None
        try:
            y = layer(src, pos, ref, shapes, lsi, None)
        finally:
            T.ENCODER_BLOCKS = True
        y.backward(go)
        return [y.detach().clone(), src.grad.clone(), pos.grad.clone()] + [p_.grad.clone() for p_ in params], type(y.grad_fn).__name__
    ours, name = run(True)
    assert "FFNBlock" in name
    ref_out, name0 = run(False)
    assert "FFNBlock" not in name0
    for a, b in zip(ours, ref_out):
        assert (a - b).abs().max() <= 3e-5 * max(b.abs().max().item(), 1e-3)


@pytest.mark.parametrize("levels", [[(47, 176), (24, 88), (12, 44), (6, 22)],          # config 4: 1408x376, S = 11044
                                    [(160, 240), (80, 120), (40, 60), (20, 30)]])       # config 5: 1920x1280, S = 51000
def test_other_baseline_geometries_encoder_shape_vs_c_oracle(levels):
    """BASELINE configs 4 and 5: the encoder-shaped call (Lq = S, queries on the pixel grid, +-4 px offsets)
    of one sample against the C oracle -- forward and all three gradients, planned kernels (host geometry attached)."""
    rng = np.random.default_rng(11)
    shapes = np.array(levels, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    B, M, D, L, P = 1, 8, 32, 4, 4
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2) for h, w in levels])
    off = rng.uniform(-4, 4, (B, S, M, L, P, 2)) / shapes[None, None, None, :, None, ::-1]
    loc = (ref[None, :, None, None, None, :] + off).astype(np.float32)
    value = rng.standard_normal((B, S, M, D)).astype(np.float32)
    w = rng.standard_normal((B, S, M, L * P))
    w = (np.exp(w) / np.exp(w).sum(-1, keepdims=True)).reshape(B, S, M, L, P).astype(np.float32)
    go = rng.standard_normal((B, S, M * D)).astype(np.float32)
    want, want64 = _oracle_want(value, shapes, lsi, loc, w, go)
    MSDA = _msda()
    s_dev = _dev(shapes)
    starts = [0]
    for h, wd in levels[:-1]:
        starts.append(starts[-1] + h * wd)
    MSDA.attach_host_geometry(s_dev, _dev(lsi), levels, starts)
    v, i, lc, ww, g = map(_dev, (value, lsi, loc, w, go))
    out = MSDA.ms_deform_attn_forward(v, s_dev, i, lc, ww, 64)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, s_dev, i, lc, ww, g, 64)
    for got, r, name in ((out, want[0], "out"), (gv, want[1], "grad_value"), (gl, want[2], "grad_loc"), (gw, want[3], "grad_attw")):
        _close(got, r, 1e-4, name)
    _close(gv, want64[1], 1e-4, "grad_value vs f64")


def test_level_embed_gradient_from_the_encoder_blocks():
    """The whole transformer (train mode, dropout 0 so both paths see the same arithmetic) with the level_embed gradient
    produced inside the encoder blocks from per-level column sums, against the path where pos carries the gradient:
    level_embed.grad, the feature gradients and a few parameter gradients."""
    import yaml
    from monosowa_amd.monodetr import depthaware_transformer as T
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "monodetr.yaml")))["model"]
    torch.manual_seed(0)
    tr = T.build_depthaware_transformer(dict(cfg, dropout=0.0)).cuda().train()
    for m in tr.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.MultiheadAttention)):
            m.p = 0.0 if isinstance(m, torch.nn.Dropout) else None
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
    tr.decoder.bbox_embed = torch.nn.ModuleList([T.MLP(256, 256, 6, 3) for _ in range(3)]).cuda()
    tr.decoder.dim_embed = torch.nn.ModuleList([T.MLP(256, 256, 3, 2) for _ in range(3)]).cuda()
    B = 16
    levels = [(24, 80), (12, 40), (6, 20), (3, 10)]
    srcs = [torch.randn(B, 256, h, w, device="cuda", requires_grad=True) for h, w in levels]
    masks = [torch.zeros(B, h, w, dtype=torch.bool, device="cuda") for h, w in levels]
    pos = [torch.randn(B, 256, h, w, device="cuda") for h, w in levels]
    query = torch.randn(550, 512, device="cuda")
    dpe = torch.randn(B, 256, *levels[0], device="cuda")
    params = [tr.level_embed, tr.encoder.layers[0].self_attn.sampling_offsets.weight, tr.encoder.layers[2].linear1.bias]

    def run(flag):
        T.LEVEL_EMBED_IN_BLOCK = flag
        for t in srcs + params:
            t.grad = None
        try:
            hs = tr(srcs, masks, pos, query, dpe, dpe, all_valid=True)[0]
        finally:
            T.LEVEL_EMBED_IN_BLOCK = True
        hs.square().mean().backward()
        return [hs.detach().clone()] + [t.grad.clone() for t in params + srcs]
    a, b = run(True), run(False)
    for x, y, n in zip(a, b, ["hs", "level_embed", "offsets.weight", "linear1.bias"] + ["src%d" % i for i in range(4)]):
        assert (x - y).abs().max() <= 1e-4 * max(y.abs().max().item(), 1e-6), n


def test_strided_fused_operator_reads_a_merged_projection_in_place():
    """msda_fused_*_strided_f32 (ABI v5): offsets | logits as column blocks of one [B, Lq, 384] buffer give the same
    output and gradients as the contiguous fused operator on the split tensors (same kernels, other addresses: bitwise but for
    grad_value's summation order)."""
    MSDA = _msda()
    torch.manual_seed(9)
    levels = [(12, 40), (6, 20), (3, 10), (2, 5)]
    B, M, D, L, P, Lq = 2, 8, 32, 4, 4, 333
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    starts = [0]
    for h, w in levels[:-1]:
        starts.append(starts[-1] + h * w)
    MSDA.attach_host_geometry(shapes, lsi, levels, starts)
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, device="cuda")
    proj = torch.randn(B, Lq, M * 48, device="cuda")
    ref = torch.rand(B, Lq, L, 2, device="cuda")
    go = torch.randn(B, Lq, M * D, device="cuda")
    off = proj[..., :M * 32].contiguous().view(B, Lq, M, L, P, 2)
    logit = proj[..., M * 32:].contiguous().view(B, Lq, M, L * P)
    out_m = MSDA.ms_deform_attn_fused_forward_merged(value, shapes, lsi, proj, ref)
    out_s = MSDA.ms_deform_attn_fused_forward(value, shapes, lsi, off, logit, ref)
    assert torch.equal(out_m, out_s)
    gv_m, gp = MSDA.ms_deform_attn_fused_backward_merged(value, shapes, lsi, proj, ref, go)
    gv_s, goff, glog = MSDA.ms_deform_attn_fused_backward(value, shapes, lsi, off, logit, ref, go)
    # grad_value: the row-band scatter sums a row's contributions in the order their slots were taken (f32, like the reference's
    # float atomics: order-dependent in the last bits) -- equal to rounding, not bitwise
    assert (gv_m - gv_s).abs().max() <= 2e-6 * gv_s.abs().max()
    assert torch.equal(gp[..., :M * 32], goff.view(B, Lq, -1)) and torch.equal(gp[..., M * 32:], glog.view(B, Lq, -1))


@pytest.mark.gpu
def test_saved_backward_from_a_plan_made_ahead_on_a_side_stream_equals_the_self_planning_one():
    """ABI v9: ``plan_saved_backward`` runs the directional statistics / plan / candidate tables right behind the forward on a side
    stream; the backward that starts from that workspace must produce exactly what the backward that plans for itself produces
    (same kernels on the same tables), with and without a padding mask, and a plan nobody uses must be harmless."""
    from monosowa_amd import MultiScaleDeformableAttention as M
    MSDA = _msda()
    torch.manual_seed(5)
    levels = [(48, 160), (24, 80), (12, 40), (6, 20)]
    B, Mh, D, L, P = 2, 8, 32, 4, 4
    shapes = torch.tensor(levels, dtype=torch.long, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    MSDA.attach_host_geometry(shapes, lsi, levels, lsi.tolist())
    S = int(shapes.prod(1).sum())
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w,
                                                indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in levels])
    ref = ref[None, :, None, :].expand(B, S, L, 2).contiguous()
    value = torch.randn(B, S, Mh, D, device="cuda")
    proj = torch.cat([torch.randn(B, S, Mh * 32, device="cuda") * 2.5, torch.randn(B, S, Mh * 16, device="cuda")], -1)
    go = torch.randn(B, S, Mh * D, device="cuda")
    for mask in (None, torch.rand(B, S, device="cuda") < 0.2):
        assert M.fused_save_supported(value, shapes, lsi, S, 2)
        out, loc, attw = M.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref, mask)
        was, M.PLAN_AHEAD = M.PLAN_AHEAD, True                           # (opt-in: see the module)
        try:
            plan = M.plan_saved_backward(value, shapes, lsi, loc)
            unused = M.plan_saved_backward(value, shapes, lsi, loc)      # dropped without a backward
        finally:
            M.PLAN_AHEAD = was
        assert plan is not None
        del unused
        torch.randn(1 << 22, device="cuda").sum()                        # main-stream work between the two calls
        gv_a, gp_a = M.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go, mask, plan=plan)
        gv_b, gp_b = M.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go, mask)
        torch.cuda.synchronize()
        assert torch.equal(gp_a, gp_b)
        # grad_value: a cell's points are summed in the order their lanes drew bucket slots (and far points / shared coarse tiles
        # with atomics): equal up to the order of float additions, call to call
        assert (gv_a - gv_b).abs().max() <= 1e-5 * gv_b.abs().max()
        # a plan is only good under the options it was made with: after msda_set_option the stale handle is ignored (its stamp no
        # longer matches) and the backward plans for itself -- same gradients, no MSDA_E_UNSUPPORTED
        from monosowa_amd import _lib
        _lib.set_option("plan_reach", 6)
        try:
            gv_c, gp_c = M.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go, mask, plan=plan)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("plan_reach", 8)
        assert torch.equal(gp_c, gp_b) and (gv_c - gv_b).abs().max() <= 1e-5 * gv_b.abs().max()
