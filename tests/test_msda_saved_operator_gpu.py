"""The TRAINING operator of the self-attention shape -- msda_fused_forward_view_f32(save) / msda_fused_backward_view_f32(saved)
behind ms_deform_attn_fused_forward_merged_save / ..._backward_merged_saved: window gather, cell scatter, device-side
directional plan, level-major saves -- against the C oracle (oracle/msda_oracle.c, cuh:33-159,237-403) through the
PyTorch-evaluated prologue (ms_deform_attn.py:146-152), at

  * the config 4 / config 5 geometries of BASELINE.json (47x176 ... S = 11,044, B = 2; 160x240 ... S = 51,000, B = 1): where the
    32-bit plane addressing and the candidate-table capacities are closest to their limits;
  * offset patterns that pile thousands of points onto ONE bilinear cell (offsets are unbounded in the reference,
    ms_deform_attn.py:145-155): the cell scatter's bucket-overflow rounds (msda_scatter_rows.hip), proven reached through the
    library's diagnostic counter;
  * the opt-in exact scan lists with a forced small list capacity: the per-point far path (msda_bin.hip).

f32: 1e-4 of each tensor's max (north_star)."""
import numpy as np
import pytest
import torch

from oracle import msda_oracle as O

pytestmark = pytest.mark.gpu

KITTI = [(48, 160), (24, 80), (12, 40), (6, 20)]            # configs[1]: 1280x384, S = 10,200
CONFIG4 = [(47, 176), (24, 88), (12, 44), (6, 22)]          # 1408x376, S = 11,044
CONFIG5 = [(160, 240), (80, 120), (40, 60), (20, 30)]       # 1920x1280, S = 51,000
M, D, L, P = 8, 32, 4, 4


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


def _pixel_centres(levels):
    return np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                           for h, w in levels]).astype(np.float32)                # [S, 2] (x, y)


def _inputs(levels, B, seed, offsets_fn):
    rng = np.random.default_rng(seed)
    shapes = np.array(levels, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    ref = _pixel_centres(levels)
    offsets = offsets_fn(rng, (B, S, M, L, P, 2)).astype(np.float32)
    logits = rng.standard_normal((B, S, M, L * P)).astype(np.float32)
    value = rng.standard_normal((B, S, M, D)).astype(np.float32)
    go = rng.standard_normal((B, S, M * D)).astype(np.float32)
    return shapes, lsi, ref, offsets, logits, value, go


def _run_saved_pair_vs_oracle(levels, shapes, lsi, ref, offsets, logits, value, go, check, set_options=()):
    """Forward (saving) + backward (saved) of the whole batch on the GPU; samples `check` against the C oracle."""
    MSDA = _msda()
    from monosowa_amd import _lib
    B, S = value.shape[0], value.shape[1]
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, [tuple(x) for x in levels], lsi.tolist())
    proj = torch.cat([_dev(offsets).reshape(B, S, M * 32), _dev(logits).reshape(B, S, M * 16)], -1).contiguous()
    refp = _dev(np.broadcast_to(ref[None, :, None, :], (B, S, 4, 2)).copy())
    v, g = _dev(value), _dev(go)
    assert MSDA.fused_save_supported(v, s, i, S), "the training pair must cover this geometry"
    lib = _lib.load()
    for name, val, _ in set_options:
        assert lib.msda_set_option(name.encode(), val) == 0
    try:
        _lib.debug_counter("scatter_overflow_rounds")                           # reset
        out, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(v, s, i, proj, refp)
        gv, gproj = MSDA.ms_deform_attn_fused_backward_merged_saved(v, s, i, loc, attw, refp, g)
        torch.cuda.synchronize()
        overflow_rounds = _lib.debug_counter("scatter_overflow_rounds")
    finally:
        for name, _, default in set_options:
            assert lib.msda_set_option(name.encode(), default) == 0
    norm = torch.stack([s[:, 1], s[:, 0]], -1).float()
    for b in check:
        off_t = _dev(offsets[b:b + 1]).requires_grad_(True)
        log_t = _dev(logits[b:b + 1]).requires_grad_(True)
        loc_t = refp[b:b + 1, :, None, :, None, :] + off_t / norm[None, None, None, :, None, :]
        aw_t = torch.softmax(log_t, -1).view(1, S, M, 4, 4)
        loc_b, aw_b = loc_t.detach().cpu().numpy(), aw_t.detach().cpu().numpy()
        want = (O.forward(value[b:b + 1], shapes, lsi, loc_b, aw_b),) + O.backward(value[b:b + 1], shapes, lsi, loc_b, aw_b, go[b:b + 1])
        _close(out[b:b + 1], want[0], 1e-4, "out[%d]" % b)
        _close(gv[b:b + 1], want[1], 1e-4, "grad_value[%d]" % b)
        g_off, g_log = torch.autograd.grad([loc_t, aw_t], [off_t, log_t], [torch.from_numpy(want[2]).cuda(), torch.from_numpy(want[3]).cuda()])
        got_off = gproj[b:b + 1, :, :M * 32].reshape(1, S, M, 4, 4, 2)
        got_log = gproj[b:b + 1, :, M * 32:].reshape(1, S, M, 16)
        assert (got_off - g_off).abs().max() <= 1e-4 * g_off.abs().max(), "grad_offsets[%d]" % b
        assert (got_log - g_log).abs().max() <= 1e-4 * g_log.abs().max(), "grad_logits[%d]" % b
    return overflow_rounds


@pytest.mark.parametrize("offsets", ["uniform4", "normal8"])
@pytest.mark.parametrize("levels,B", [(CONFIG4, 2), (CONFIG5, 1)], ids=["config4", "config5"])
def test_fused_saved_operator_at_config4_and_config5_geometries_vs_c_oracle(levels, B, offsets):
    """Forward, grad_value and grad_proj of the pair the train step runs, at the two other BASELINE geometries, with the module's
    +-4 px range and with N(0, 8 px) offsets (three quarters of the points outside every LDS window, half beyond the scan bounds)."""
    fn = (lambda rng, shp: rng.uniform(-4, 4, shp)) if offsets == "uniform4" else (lambda rng, shp: 8.0 * rng.standard_normal(shp))
    data = _inputs(levels, B, 17 + B, fn)
    _run_saved_pair_vs_oracle(levels, *data, check=range(B))


def _row_tiling(H, W, cells=128, mg=8):
    """Tile extents of the cell scatter at one level (msda_scatter_plan.h: make_row_plan) -- only used to aim at a tile corner."""
    if (H + 1) * (W + 1) <= cells:
        return H, W
    best, th, tw = None, H, W
    for nty in range(1, H + 1):
        h = (H + nty - 1) // nty
        w_max = cells // (h + 1) - 1
        if w_max < 1:
            continue
        ntx = (W + w_max - 1) // w_max
        w = (W + ntx - 1) // ntx
        cost = nty * ntx * (h + mg) * (w + mg)
        if best is None or cost < best:
            best, th, tw = cost, h, w
    n_tx = (W + tw - 1) // tw
    n_ty = (H + th - 1) // th
    return (H + n_ty - 1) // n_ty, (W + n_tx - 1) // n_tx


def _pile_on_cell(rng, offsets, ref, levels, lsi, level, cell_y, cell_x, radius):
    """Every query (of any level) whose own pixel lies within `radius` pixels (of `level`) of the cell sends all four points it has
    at `level` into the bilinear cell (cell_y, cell_x): h_low = cell_y, w_low = cell_x, random fractions.  Returns the count."""
    H, W = levels[level]
    qx, qy = ref[:, 0] * W - 0.5, ref[:, 1] * H - 0.5                            # the queries' centres in pixels of `level`
    sel = np.nonzero((np.abs(np.floor(qx) - cell_x) <= radius) & (np.abs(np.floor(qy) - cell_y) <= radius))[0]
    B = offsets.shape[0]
    fx = rng.uniform(0.1, 0.9, (B, sel.size, M, P)).astype(np.float32)
    fy = rng.uniform(0.1, 0.9, (B, sel.size, M, P)).astype(np.float32)
    # loc * W - 0.5 = cell_x + fx   with   loc = ref + offset / W
    new = np.stack([(cell_x + fx + 0.5) - ref[sel, 0][None, :, None, None] * W,
                    (cell_y + fy + 0.5) - ref[sel, 1][None, :, None, None] * H], -1)          # [B, n, M, P, 2]
    lvl = offsets[:, :, :, level]                                                # view [B, S, M, P, 2]
    lvl[:, sel] = new
    return sel.size


def test_cell_scatter_bucket_overflow_rounds_vs_c_oracle():
    """Thousands of neighbouring queries put all their level-3 points on ONE cell of level 3, and all their level-0 points on the
    cell at a level-0 tile's top-left corner (its apron: accumulated by four tiles): a batch of 256 candidates brings up to 1,024
    points to a bucket of 32 -- the `again` rounds of scatter_rows_kernel, dozens per batch.  Against the C oracle, f32 1e-4."""
    B = 2
    shapes, lsi, ref, offsets, logits, value, go = _inputs(KITTI, B, 301, lambda rng, shp: rng.uniform(-3, 3, shp))
    rng = np.random.default_rng(302)
    n3 = _pile_on_cell(rng, offsets, ref, KITTI, lsi, 3, 3, 10, 5)
    th0, tw0 = _row_tiling(*KITTI[0])
    # the cell above-left of the second tile's first output row / column: on the apron of four level-0 tiles
    n0 = _pile_on_cell(rng, offsets, ref, KITTI, lsi, 0, th0 - 1, 2 * tw0 - 1, 5)
    assert n3 >= 3000 and n0 >= 100, (n3, n0)
    rounds = _run_saved_pair_vs_oracle(KITTI, shapes, lsi, ref, offsets, logits, value, go, check=range(B))
    assert rounds >= 100, "the overflow rounds were not reached (%d): the test no longer exercises them" % rounds


def test_exact_scan_lists_with_a_forced_small_capacity_vs_c_oracle():
    """Opt-in list-driven scatter (msda_bin.hip) with every list capped at 64 entries: most units overflow their tile's list and go
    to the gather kernel's row-atomic path as WHOLE points; a point whose cell lies on an apron (listed by two tiles, only one of
    them full) must not be added twice."""
    B = 2
    data = _inputs(KITTI, B, 311, lambda rng, shp: rng.uniform(-4, 4, shp))
    _run_saved_pair_vs_oracle(KITTI, *data, check=range(B), set_options=(("scatter_lists", 1, 0), ("scatter_lists_cap", 64, 0)))


@pytest.mark.parametrize("levels,B", [(KITTI, 16), (CONFIG4, 16), (CONFIG5, 4)], ids=["config2-B16", "config4-B16", "config5-B4"])
def test_fused_saved_pair_full_size_properties(levels, B):
    """The training pair at the batch sizes bench.py runs (BASELINE configs 2, 4, 5), where the C oracle would take minutes: properties
    that do not depend on the size.  (1) the forward is linear in value; (2) <grad_out, J v> == <J^T grad_out, v>: grad_value is the
    adjoint of the forward's value path (sums in float64); (3) the logits' gradient against a central difference of <grad_out, out>
    along a random direction (the forward is smooth in the logits; the offsets' kinks at pixel borders are left to the oracle tests)."""
    MSDA = _msda()
    gen = torch.Generator(device="cuda").manual_seed(1000 + B + levels[0][0])
    S = sum(h * w for h, w in levels)
    shapes = np.array(levels, dtype=np.int64)
    lsi = O.level_start_index(shapes)
    s, i = _dev(shapes), _dev(lsi)
    MSDA.attach_host_geometry(s, i, [tuple(x) for x in levels], lsi.tolist())
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=gen)
    offsets = (torch.rand(B, S, M * 32, device="cuda", generator=gen) * 8 - 4)
    logits = rnd(B, S, M * 16)
    refp = _dev(np.broadcast_to(_pixel_centres(levels)[None, :, None, :], (B, S, 4, 2)).copy())
    v1, v2, go = rnd(B, S, M, D), rnd(B, S, M, D), rnd(B, S, M * D)
    assert MSDA.fused_save_supported(v1, s, i, S)
    proj = torch.cat([offsets, logits], -1).contiguous()
    fwd = lambda v, pj: MSDA.ms_deform_attn_fused_forward_merged_save(v, s, i, pj, refp)
    o1, loc, attw = fwd(v1, proj)
    o2 = fwd(v2, proj)[0]
    o12 = fwd(v1 + 2 * v2, proj)[0]
    assert (o12 - (o1 + 2 * o2)).abs().max() <= 1e-4 * o12.abs().max(), "linearity in value"
    gv, gproj = MSDA.ms_deform_attn_fused_backward_merged_saved(v1, s, i, loc, attw, refp, go)
    lhs = (go.double() * o1.reshape(B, S, M * D).double()).sum()
    rhs = (gv.double() * v1.double()).sum()
    assert abs(lhs - rhs) <= 1e-6 * abs(lhs), "adjoint identity of the value path: %r vs %r" % (lhs.item(), rhs.item())
    direction = rnd(B, S, M * 16)
    eps = 1e-2
    dot = lambda pj: (go.double() * fwd(v1, pj)[0].reshape(B, S, M * D).double()).sum()
    plus = dot(torch.cat([offsets, logits + eps * direction], -1).contiguous())
    minus = dot(torch.cat([offsets, logits - eps * direction], -1).contiguous())
    numeric = (plus - minus) / (2 * eps)
    analytic = (gproj[:, :, M * 32:].double() * direction.double()).sum()
    scale = (gproj[:, :, M * 32:].double().abs() * direction.double().abs()).sum()        # the size of the terms that cancel in `analytic`
    print("adjoint rel %.2e; logit gradient: |numeric - analytic| / scale = %.2e" % (abs(lhs - rhs) / abs(lhs), abs(numeric - analytic) / scale))
    assert abs(numeric - analytic) <= 1e-6 * scale, "logit gradient vs central difference: %r vs %r (scale %r)" % (numeric.item(), analytic.item(), scale.item())
