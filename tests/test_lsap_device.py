"""The matcher's assignment solver on the device (csrc/lsap_device.hip, C-ABI mono_lsap_match_flat_f32) against the host
solver (csrc/lsap.cpp) and scipy.optimize.linear_sum_assignment -- the reference's solver (matcher.py:94-103): the same
pairs, bit for bit, including tied cost matrices, both orientations (fewer / more targets than queries per group), images
without targets and the training shape (3 layers x 16 images x 11 groups of 50 queries)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rng, NL, B, Q, sizes, kind, G):
    N = max(max(sizes), 1)
    if kind == "random":
        c = rng.standard_normal((NL, B, Q, N)).astype(np.float32)
    elif kind == "ties":                      # few distinct values: most minima are tied
        c = rng.integers(0, 4, (NL, B, Q, N)).astype(np.float32)
    elif kind == "constant":
        c = np.ones((NL, B, Q, N), np.float32)
    elif kind == "matcher_like":              # clustered costs with exact duplicates (repeated targets / queries)
        base = rng.standard_normal((NL, B, Q // 5 + 1, N)).astype(np.float32)
        c = np.repeat(base, 5, axis=2)[:, :, :Q] + (rng.integers(0, 2, (NL, B, Q, N)) * 0.5).astype(np.float32)
    elif kind == "with_inf":                  # +inf entries are legal as long as an assignment exists
        c = rng.standard_normal((NL, B, Q, N)).astype(np.float32)
        c[rng.random(c.shape) < 0.2] = np.inf
        gq = Q // G
        for g in range(G):                    # a finite diagonal per group keeps every problem feasible
            for d in range(min(gq, N)):
                c[:, :, g * gq + d, d] = rng.standard_normal((NL, B)).astype(np.float32)
    else:
        raise ValueError(kind)
    return c


def _host(c, sizes, G):
    from monosowa_amd import lsap
    return lsap.match_flat(c, np.asarray(sizes, np.int64), G, padded=True)


def _device(c, sizes, G):
    from monosowa_amd import pointwise
    blocks = torch.from_numpy(c).cuda()
    assert pointwise.device_lsap_supported(blocks, sizes, G)
    status = torch.zeros((), dtype=torch.int32, device="cuda")
    idx = pointwise.device_lsap_match_flat(blocks, sizes, G, status)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), int(status.item())


@pytest.mark.parametrize("kind", ["random", "ties", "constant", "matcher_like", "with_inf"])
@pytest.mark.parametrize("NL,B,Q,G,sizes", [
    (3, 16, 550, 11, [7, 12, 0, 50, 1, 33, 49, 50, 2, 18, 25, 5, 50, 9, 14, 21]),      # the training shape
    (1, 3, 50, 1, [3, 50, 17]),                                                      # evaluation: one group
    (2, 2, 24, 3, [20, 8]),                                                          # more targets than queries per group (8 queries)
    (1, 2, 128, 1, [64, 5]),                                                         # two list positions per lane
    (1, 1, 6, 2, [3]),
])
def test_device_assignments_equal_the_host_solver(NL, B, Q, G, sizes, kind):
    rng = np.random.default_rng(NL * 1000003 + B * 10007 + Q * 101 + G + len(kind))
    c = _case(rng, NL, B, Q, sizes, kind, G)
    want = _host(c, sizes, G)
    got, status = _device(c, sizes, G)
    assert status == 0
    assert got.shape == want.shape and got.dtype == np.int64
    assert np.array_equal(got, want)


def test_device_assignments_equal_scipy_per_problem():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    NL, B, Q, G, sizes = 2, 4, 100, 4, [30, 7, 25, 11]
    c = rng.integers(0, 6, (NL, B, Q, max(sizes))).astype(np.float32)              # heavy ties
    got, status = _device(c, sizes, G)
    assert status == 0
    gq, pos, toff = Q // G, 0, np.concatenate([[0], np.cumsum(sizes)[:-1]])
    K = got.shape[2]
    for l in range(NL):
        pos = 0
        for b in range(B):
            for g in range(G):
                r, t = linear_sum_assignment(c[l, b, g * gq:(g + 1) * gq, :sizes[b]].astype(np.float64))
                k = len(r)
                assert np.array_equal(got[0, l, pos:pos + k], np.full(k, b))
                assert np.array_equal(got[1, l, pos:pos + k], r + g * gq)
                assert np.array_equal(got[2, l, pos:pos + k], t + toff[b])
                pos += k
        assert pos == K


def test_invalid_costs_raise_the_flag_and_leave_usable_indices():
    """scipy raises on NaN / -inf entries and on infeasible matrices; the device solver cannot raise from a kernel: it sets
    bit 0 of the status word (read by the criterion without stalling the step) and writes in-range indices."""
    rng = np.random.default_rng(9)
    NL, B, Q, G, sizes = 1, 2, 20, 2, [4, 6]
    for poison in (np.nan, -np.inf, "row_of_inf"):
        c = rng.standard_normal((NL, B, Q, max(sizes))).astype(np.float32)
        if poison == "row_of_inf":
            c[0, 1, :, 2] = np.inf                  # target 2 of image 1 is unreachable from every query: infeasible
        else:
            c[0, 1, 3, 2] = poison
        got, status = _device(c, sizes, G)
        assert status & 1
        assert got[0].min() >= 0 and got[0].max() < B and got[1].min() >= 0 and got[1].max() < Q
        assert got[2].min() >= 0 and got[2].max() < sum(sizes)
        from monosowa_amd import lsap
        with pytest.raises(ValueError):
            lsap.match_flat(c, np.asarray(sizes, np.int64), G, padded=True)


def test_criterion_with_the_device_matcher_equals_the_host_matcher():
    """The whole criterion (fused tail) on the training shape: losses and the gradient w.r.t. the predictions with
    matcher.DEVICE_LSAP on and off are identical (the same pairs feed the same kernels)."""
    import os
    import yaml
    from monosowa_amd.helpers.model_helper import build_model
    from monosowa_amd.monodetr import matcher
    from monosowa_amd.synthetic import make_batch, prepare_targets
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "monodetr.yaml")))
    torch.manual_seed(3)
    _, crit = build_model(dict(cfg["model"], device="cuda"))
    crit = crit.cuda().train()
    B, Q = 8, 550
    _, _, targets, _ = make_batch(B, "cuda", seed=11, resolution=(640, 192))
    w, h = crit.depth_map_size if hasattr(crit, "depth_map_size") else (80, 24)

    def outputs(seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        mk = lambda *s: torch.randn(*s, device="cuda", generator=g).requires_grad_(True)
        layer = lambda: {"pred_logits": mk(B, Q, 3), "pred_boxes": torch.sigmoid(mk(B, Q, 6)), "pred_3d_dim": mk(B, Q, 3).abs() + 0.5,
                         "pred_depth": mk(B, Q, 2), "pred_angle": mk(B, Q, 24)}
        out = layer()
        out["aux_outputs"] = [layer(), layer()]
        out["pred_depth_map_logits"] = mk(B, 81, h, w)
        return out
    res = []
    for on in (True, False):
        saved = matcher.DEVICE_LSAP
        matcher.DEVICE_LSAP = on
        try:
            out = outputs(17)
            tl = prepare_targets(targets, B)
            ld = crit(out, tl)
            total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
            leaves = [out["pred_logits"], out["aux_outputs"][0]["pred_logits"], out["aux_outputs"][1]["pred_depth"]]
            grads = torch.autograd.grad(total, leaves)
            res.append(({k: float(v.detach()) for k, v in ld.items()}, [g.clone() for g in grads]))
        finally:
            matcher.DEVICE_LSAP = saved
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_shapes_beyond_the_kernels_tables_stay_on_the_host_solver():
    """More than 128 queries per group (or targets per image): ``device_lsap_supported`` says no and the matcher takes the host
    path (one copy, one wait) -- same contract, same pairs as scipy."""
    import os
    import yaml
    from scipy.optimize import linear_sum_assignment
    from monosowa_amd import pointwise
    from monosowa_amd.monodetr import matcher as M
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = M.build_matcher(yaml.safe_load(open(os.path.join(root, "configs", "monodetr.yaml")))["model"])
    g = torch.Generator(device="cuda").manual_seed(2)
    NL, B, Q, G, sizes = 1, 2, 300, 1, [5, 9]
    logits = torch.randn(NL, B, Q, 3, device="cuda", generator=g)
    boxes = torch.rand(NL, B, Q, 6, device="cuda", generator=g) * 0.4 + 0.1
    T = sum(sizes)
    flat = {"labels": torch.randint(0, 3, (T,), device="cuda", generator=g), "boxes_3d": torch.rand(T, 6, device="cuda", generator=g) * 0.4 + 0.1}
    handle = m.match_layers_begin(logits, boxes, flat, sizes, G)
    assert not isinstance(handle[0][0], str)                      # not the device solver's handle
    assert not pointwise.device_lsap_supported(torch.empty(NL, B, Q, max(sizes), device="cuda"), sizes, G)
    idx = m.match_layers_end_flat(handle)
    idx = idx.cpu().numpy() if torch.is_tensor(idx) else np.asarray(idx)
    blocks = handle[0][0].numpy()
    pos, toff = 0, 0
    for b in range(B):
        r, t = linear_sum_assignment(blocks[0, b, :, :sizes[b]].astype(np.float64))
        assert np.array_equal(idx[1, 0, pos:pos + len(r)], r) and np.array_equal(idx[2, 0, pos:pos + len(r)], t + toff)
        pos += len(r)
        toff += sizes[b]
