"""Host-side helpers: the AdamW variant (bit-for-bit against a literal per-parameter restatement of
lib/helpers/optimizer_helper.py:69-129), checkpoint dictionary, LR schedule, detection bookkeeping
(decode_helper.py:58-111 -- top-k / // / % / gather must be exact), KITTI result writer, synthetic data
contract, the end-to-end CPU plumbing step (BASELINE configs[0], reduced resolution)."""
import math
import os

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return yaml.safe_load(open(os.path.join(ROOT, "configs", "monodetr.yaml")))


def _reference_adamw_step(params, grads, state, lr, wd, step, betas=(0.9, 0.999), eps=1e-8):
    """Literal per-parameter loop of the reference update."""
    b1, b2 = betas
    for p, g, st in zip(params, grads, state):
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = st["v"].sqrt().add_(eps)
        step_size = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        p.add_(torch.mul(p, wd).addcdiv_(st["m"], denom, value=1), alpha=-step_size)


def test_adamw_matches_reference_update_bitwise():
    from monosowa_amd.helpers.optimizer_helper import AdamW
    torch.manual_seed(0)
    shapes = [(7, 5), (5,), (3, 4, 2), (1,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref = [p.detach().clone() for p in ps]
    st = [{"m": torch.zeros_like(p), "v": torch.zeros_like(p)} for p in ref]
    opt = AdamW([{"params": ps[:2], "weight_decay": 0}, {"params": ps[2:], "weight_decay": 1e-4}], lr=2e-4)
    for step in range(1, 6):
        gs = [torch.randn_like(p) for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
        _reference_adamw_step(ref[:2], gs[:2], st[:2], 2e-4, 0, step)
        _reference_adamw_step(ref[2:], gs[2:], st[2:], 2e-4, 1e-4, step)
        for p, r in zip(ps, ref):
            assert torch.equal(p.detach(), r)
    sd = opt.state_dict()["state"][0]
    assert set(sd.keys()) == {"step", "exp_avg", "exp_avg_sq"}          # reference state keys


def test_build_optimizer_groups_biases_without_decay():
    from monosowa_amd.helpers.optimizer_helper import build_optimizer
    m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.LayerNorm(4))
    opt = build_optimizer({"type": "adamw", "lr": 1e-3, "weight_decay": 0.1}, m)
    assert opt.param_groups[0]["weight_decay"] == 0 and len(opt.param_groups[0]["params"]) == 2
    assert opt.param_groups[1]["weight_decay"] == 0.1 and len(opt.param_groups[1]["params"]) == 2
    with pytest.raises(NotImplementedError):
        build_optimizer({"type": "lion", "lr": 1e-3, "weight_decay": 0.1}, m)


def test_lr_schedule_and_checkpoint_roundtrip(tmp_path):
    from monosowa_amd.helpers.optimizer_helper import build_optimizer
    from monosowa_amd.helpers.save_helper import get_checkpoint_state, load_checkpoint, save_checkpoint
    from monosowa_amd.helpers.scheduler_helper import build_lr_scheduler
    m = torch.nn.Linear(3, 2)
    opt = build_optimizer({"type": "adamw", "lr": 2e-4, "weight_decay": 1e-4}, m)
    sched, warm = build_lr_scheduler({"warmup": False, "decay_rate": 0.1, "decay_list": [2, 4]}, opt, last_epoch=-1)
    assert warm is None
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert np.allclose(lrs, [2e-4, 2e-4, 2e-5, 2e-5, 2e-6])
    _, warm = build_lr_scheduler({"warmup": True, "decay_rate": 0.1, "decay_list": [2]}, opt, last_epoch=-1)
    assert warm is not None
    m(torch.randn(1, 3)).sum().backward()
    opt.step()
    state = get_checkpoint_state(m, opt, epoch=3, best_result=1.5, best_epoch=2)
    assert set(state) == {"epoch", "model_state", "optimizer_state", "best_result", "best_epoch"}
    save_checkpoint(state, str(tmp_path / "checkpoint_epoch_3"))
    m2 = torch.nn.Linear(3, 2)
    opt2 = build_optimizer({"type": "adamw", "lr": 2e-4, "weight_decay": 1e-4}, m2)
    ep, best, best_ep = load_checkpoint(m2, opt2, str(tmp_path / "checkpoint_epoch_3.pth"), "cpu")
    assert (ep, best, best_ep) == (3, 1.5, 2)
    assert torch.equal(m2.weight, m.weight)
    with pytest.raises(FileNotFoundError):
        load_checkpoint(m2, None, str(tmp_path / "nope.pth"), "cpu")


def test_extract_dets_bookkeeping_is_exact():
    from monosowa_amd.helpers.decode_helper import extract_dets_from_outputs
    torch.manual_seed(1)
    B, Q, C = 3, 50, 3
    out = {"pred_logits": torch.randn(B, Q, C), "pred_boxes": torch.rand(B, Q, 6) * 0.2 + 0.3,
           "pred_angle": torch.randn(B, Q, 24), "pred_3d_dim": torch.randn(B, Q, 3), "pred_depth": torch.randn(B, Q, 2)}
    dets = extract_dets_from_outputs(out, K=50, topk=50)
    assert dets.shape == (B, 50, 37)
    prob = out["pred_logits"].sigmoid().numpy().reshape(B, -1)
    for b in range(B):
        order = np.argsort(-prob[b], kind="stable")[:50]
        assert np.array_equal(dets[b, :, 0].numpy().astype(np.int64), order % C)             # labels
        assert np.array_equal(dets[b, :, 1].numpy(), prob[b][order])                         # scores, exact
        qi = order // C
        assert np.array_equal(dets[b, :, 6].numpy(), out["pred_depth"][b, qi, 0].numpy())    # gathered depth
        assert np.array_equal(dets[b, :, 7:31].numpy(), out["pred_angle"][b, qi].numpy())
        assert np.array_equal(dets[b, :, 34].numpy(), out["pred_boxes"][b, qi, 0].numpy())   # x3d
        assert np.allclose(dets[b, :, 36].numpy(), np.exp(-out["pred_depth"][b, qi, 1].numpy()))


def test_decode_and_kitti_writer(tmp_path):
    from monosowa_amd.helpers.decode_helper import PinholeCalib, class2angle, decode_detections, get_heading_angle
    assert class2angle(0, 0.1) == pytest.approx(0.1)
    assert class2angle(11, 0.2, to_label_format=True) == pytest.approx(11 * math.pi / 6 + 0.2 - 2 * math.pi)
    heading = np.zeros(24, np.float32)
    heading[3] = 5.0
    heading[15] = 0.25
    assert get_heading_angle(heading) == pytest.approx(3 * math.pi / 6 + 0.25)
    P2 = np.array([[700., 0, 600., 40.], [0, 700., 180., 2.], [0, 0, 1, 0.003]])
    cal = PinholeCalib(P2)
    xyz = cal.img_to_rect(np.array([650.]), np.array([200.]), np.array([10.]))
    assert np.allclose(xyz, [[(650 - 600) * 10 / 700 + 40 / -700, (200 - 180) * 10 / 700 + 2 / -700, 10.]])
    dets = np.zeros((1, 2, 37), np.float32)
    dets[0, 0, :7] = [1, 0.9, 0.5, 0.5, 0.1, 0.2, 20.0]
    dets[0, 0, 31:34] = [1.5, 1.6, 3.9]
    dets[0, 0, 34:37] = [0.5, 0.5, 0.8]
    dets[0, 1, 1] = 0.1                                   # below threshold
    info = {"img_size": np.array([[1242, 375]]), "height_crop": np.array([1.0]), "canonical_scale": np.array([0.5]),
            "img_id": np.array([7])}
    res = decode_detections(dets, info, [cal], np.zeros((3, 3), np.float32), threshold=0.2)
    assert list(res) == [7] and len(res[7]) == 1
    p = res[7][0]
    assert p[0] == 1 and p[-1] == pytest.approx(0.9 * 0.8) and len(p) == 14
    assert p[2:6] == pytest.approx([0.5 * 1242 - 62.1, 0.5 * 375 - 37.5, 0.5 * 1242 + 62.1, 0.5 * 375 + 37.5], rel=1e-5)
    assert p[11] == pytest.approx(40.0)                   # depth leaves Canonical Object Space: 20 / 0.5

    from monosowa_amd.helpers.tester_helper import Tester

    class _DS:
        max_objs, class_name, cls_mean_size = 50, ["Pedestrian", "Car", "Cyclist"], np.zeros((3, 3))

    class _DL:
        dataset = _DS()

    import logging
    t = Tester({"type": "KITTI", "topk": 50}, None, _DL(), logging.getLogger("t"), {"save_path": str(tmp_path) + "/"}, "m")
    t.output_dir = str(tmp_path)
    t.save_results(res)
    line = open(tmp_path / "outputs" / "data" / "000007.txt").read().strip().split(" ")
    assert line[0] == "Car" and line[1:3] == ["0.0", "0"] and len(line) == 16 and line[-1] == "0.72"
    assert all(len(x.split(".")[1]) == 2 for x in line[3:])                         # '{:.2f}' everywhere


def test_synthetic_batch_contract():
    from monosowa_amd.synthetic import SyntheticKITTI, make_batch, prepare_targets
    inputs, calibs, targets, info = make_batch(3, "cpu", seed=1, resolution=(320, 96))
    assert inputs.shape == (3, 3, 96, 320) and inputs.dtype == torch.float32 and calibs.shape == (3, 3, 4)
    assert set(targets) == {"calibs", "indices", "img_size", "labels", "boxes", "boxes_3d", "depth", "size_2d", "size_3d",
                            "src_size_3d", "heading_bin", "heading_res", "mask_2d"}
    assert targets["labels"].dtype == torch.int8 and targets["heading_bin"].dtype == torch.int64
    assert targets["mask_2d"].dtype == torch.bool and targets["boxes_3d"].shape == (3, 50, 6)
    tl = prepare_targets(targets, 3)
    for b, t in enumerate(tl):
        n = int(targets["mask_2d"][b].sum())
        assert 1 <= n <= 10 and t["boxes_3d"].shape == (n, 6) and t["calibs"].shape == (n, 3, 4)
        x0 = t["boxes_3d"][:, 0] - t["boxes_3d"][:, 2]
        assert torch.allclose(t["boxes"][:, 0] - t["boxes"][:, 2] / 2, x0, atol=1e-6)
    ds = SyntheticKITTI("train", {"resolution": (320, 96), "num_samples": 4})
    a, b = ds[1], ds[1]
    assert len(ds) == 4 and np.array_equal(a[0], b[0]) and a[3]["img_id"] == 1


def test_cpu_plumbing_train_and_eval_step(monkeypatch):
    """BASELINE configs[0] (CPU-only path, random images) at reduced resolution: dict keys / shapes /
    finite loss, one optimizer step, then eval mode with 50 queries.  MSDA goes through the oracle port
    (tests only)."""
    import monosowa_amd.ms_deform_attn_func as F
    from oracle import msda_oracle as O
    from monosowa_amd.helpers.model_helper import build_model
    from monosowa_amd.helpers.optimizer_helper import build_optimizer
    from monosowa_amd.helpers.decode_helper import extract_dets_from_outputs
    from monosowa_amd.synthetic import make_batch, prepare_targets

    class _Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return O.msda_core_torch(value, shapes, loc, w)
    monkeypatch.setattr(F, "MSDeformAttnFunction", _Fn)
    cfg = _cfg()
    torch.manual_seed(cfg["random_seed"])
    mcfg = dict(cfg["model"], device="cpu", depth_map_size=(20, 6))
    model, crit = build_model(mcfg)
    opt = build_optimizer(cfg["optimizer"], model)
    inputs, calibs, targets, info = make_batch(2, "cpu", resolution=(320, 96))
    tl = prepare_targets(targets, 2)
    model.train()
    crit.train()
    out = model(inputs, calibs, tl, targets["img_size"])
    assert out["pred_logits"].shape == (2, 550, 3) and out["pred_boxes"].shape == (2, 550, 6)
    assert out["pred_3d_dim"].shape == (2, 550, 3) and out["pred_depth"].shape == (2, 550, 2)
    assert out["pred_angle"].shape == (2, 550, 24) and out["pred_depth_map_logits"].shape == (2, 81, 6, 20)
    assert len(out["aux_outputs"]) == 2
    losses = crit(out, tl)
    wd = crit.weight_dict
    expected = {"loss_ce", "loss_bbox", "loss_giou", "loss_center", "loss_depth", "loss_dim", "loss_angle", "loss_depth_map",
                "loss_tfl", "loss_mask"}
    assert expected <= set(losses) and {k + "_0" for k in expected - {"loss_depth_map"}} <= set(losses)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    assert torch.isfinite(total)
    before = model.class_embed[0].weight.detach().clone()
    total.backward()
    opt.step()
    assert not torch.equal(before, model.class_embed[0].weight)
    no_grad = sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is None)
    assert no_grad == sorted(model.unused_parameter_names())
    model.eval()
    crit.eval()
    with torch.no_grad():
        out = model(inputs, calibs, None, targets["img_size"])
    assert out["pred_logits"].shape == (2, 50, 3)
    assert extract_dets_from_outputs(out, K=50, topk=50).shape == (2, 50, 37)
    assert len(crit(out, tl)) > 0


def test_unshipped_switches_fail_loudly():
    from monosowa_amd.helpers.model_helper import build_model
    cfg = _cfg()["model"]
    for key in ("two_stage", "use_dab", "two_stage_dino", "use_dn", "use_tfl", "pretrained"):
        bad = dict(cfg, device="cpu")
        bad[key] = True
        with pytest.raises((NotImplementedError, RuntimeError)):
            build_model(bad)


def test_fast_criterion_equals_layerwise_formulation():
    """The batched criterion (one matching pass, flat indices, device rasterisation) against the
    layer-by-layer / image-by-image formulation restated from the reference, losses and gradients."""
    from monosowa_amd.monodetr import build_weight_dict
    from monosowa_amd.monodetr.criterion import SetCriterion
    from monosowa_amd.monodetr.matcher import build_matcher
    from monosowa_amd.synthetic import make_batch, prepare_targets
    cfg = _cfg()["model"]
    torch.manual_seed(5)
    B, G, Qg = 3, 11, 50
    _, _, targets, _ = make_batch(B, "cpu", seed=9, resolution=(320, 96))
    tl = prepare_targets(targets, B)
    tl[1] = {k: v[:0] for k, v in tl[1].items()}                       # an image without objects

    def outputs(Q, seed):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: torch.randn(*s, generator=g)
        layer = lambda: {"pred_logits": mk(B, Q, 3).requires_grad_(), "pred_boxes": (torch.rand(B, Q, 6, generator=g) * 0.3 + 0.2).requires_grad_(),
                         "pred_3d_dim": mk(B, Q, 3).requires_grad_(), "pred_depth": mk(B, Q, 2).requires_grad_(),
                         "pred_angle": mk(B, Q, 24).requires_grad_()}
        out = layer()
        out["pred_depth_map_logits"] = mk(B, 81, 6, 20).requires_grad_()
        out["aux_outputs"] = [layer(), layer()]
        return out

    for training in (True, False):
        Q = G * Qg if training else Qg
        res = {}
        for fast in (False, True):
            crit = SetCriterion(3, build_matcher(cfg), build_weight_dict(cfg), 0.25,
                                ["labels", "boxes", "cardinality", "depths", "dims", "angles", "center", "depth_map", "tfl"],
                                cfg=cfg, depth_map_size=(20, 6), fast=fast).train(training)
            out = outputs(Q, 3)
            ld = crit(out, tl)
            literal = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
            from monosowa_amd.monodetr.criterion import weighted_total
            total = weighted_total(ld, crit.weight_dict)          # LossDict fast path for fast=True, stack + dot otherwise
            assert torch.allclose(total, literal, rtol=1e-5), (total, literal)
            total.backward()
            res[fast] = (ld, out)
        slow, fastd = res[False][0], res[True][0]
        assert set(slow) == set(fastd), set(slow) ^ set(fastd)
        for k in slow:
            assert torch.allclose(slow[k], fastd[k], rtol=2e-5, atol=1e-6), (k, slow[k], fastd[k])
        for key in ("pred_logits", "pred_boxes", "pred_depth", "pred_3d_dim", "pred_angle", "pred_depth_map_logits"):
            assert torch.allclose(res[False][1][key].grad, res[True][1][key].grad, rtol=1e-4, atol=1e-7), key
        for a, b in zip(res[False][1]["aux_outputs"], res[True][1]["aux_outputs"]):
            assert torch.allclose(a["pred_boxes"].grad, b["pred_boxes"].grad, rtol=1e-4, atol=1e-7)


def test_rasterize_boxes_matches_painting_loop(golden_dir=None):
    from monosowa_amd.monodetr import losses as L
    g = np.load(os.path.join(ROOT, "tests", "golden", "losses.npz"))
    boxes, depth, num_gt = torch.from_numpy(g["boxes"]), torch.from_numpy(g["depth"]), [int(x) for x in g["num_gt"]]
    pad = torch.zeros(2, 3, 4)
    dpad = torch.zeros(2, 3)
    valid = torch.zeros(2, 3, dtype=torch.bool)
    pad[0, :3], pad[1, :2] = boxes[:3], boxes[3:]
    dpad[0, :3], dpad[1, :2] = depth[:3], depth[3:]
    valid[0, :3], valid[1, :2] = True, True
    dm, fg = L.rasterize_boxes(L._int_boxes(pad.view(-1, 4)).view(2, 3, 4), dpad, valid, 24, 80)
    assert torch.equal(dm, torch.from_numpy(g["depth_map"]))                  # incl. the box poking out on the left
    loss = L.DDNLoss().forward_padded(torch.from_numpy(g["depth_logits"]), pad, dpad, valid)
    assert torch.allclose(loss, torch.as_tensor(g["ddn_loss"]), rtol=1e-5)


def test_native_lsap_equals_scipy_including_ties():
    from scipy.optimize import linear_sum_assignment as ref
    from monosowa_amd import lsap
    assert lsap.available()
    rng = np.random.default_rng(0)
    for trial in range(400):
        nr, nc = int(rng.integers(1, 12)), int(rng.integers(1, 60))
        if trial % 2:
            nr, nc = nc, nr
        c = rng.integers(0, 4, (nr, nc)).astype(np.float64) if trial % 3 == 0 else rng.standard_normal((nr, nc))
        a, b = ref(c)
        x, y = lsap.linear_sum_assignment(c)
        assert np.array_equal(a, x) and np.array_equal(b, y)
    with pytest.raises(ValueError):
        lsap.linear_sum_assignment(np.full((2, 2), np.nan))
    # the flat, threaded entry point returns the same assignments as the per-image lists
    NL, B, Q, G = 3, 5, 44, 11
    sizes = np.array([3, 0, 7, 1, 6])
    cost = rng.standard_normal((NL, B, Q, 7)).astype(np.float32)
    lists = lsap.match_groups(cost, sizes, G, padded=True)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    want = np.stack([np.stack([np.concatenate([np.full(len(s_), b) for b, (s_, _) in enumerate(layer)]) for layer in lists]),
                     np.stack([np.concatenate([s_ for s_, _ in layer]) for layer in lists]),
                     np.stack([np.concatenate([t_ + offs[b] for b, (_, t_) in enumerate(layer)]) for layer in lists])])
    for threads in (1, 3):
        assert np.array_equal(lsap.match_flat(cost, sizes, G, padded=True, n_threads=threads), want)


def test_vectorised_decode_equals_loop_form():
    from monosowa_amd.helpers.decode_helper import PinholeCalib, decode_detections, decode_detections_loop
    rng = np.random.default_rng(3)
    B, K = 4, 50
    dets = rng.standard_normal((B, K, 37)).astype(np.float32)
    dets[:, :, 0] = rng.integers(0, 3, (B, K))
    dets[:, :, 1] = rng.uniform(0, 1, (B, K))
    dets[:, :, 2:6] = rng.uniform(0.05, 0.9, (B, K, 4))
    dets[:, :, 6] = rng.uniform(2, 60, (B, K))
    dets[:, :, 34:36] = rng.uniform(0.1, 0.9, (B, K, 2))
    dets[:, :, 36] = rng.uniform(0.2, 1, (B, K))
    info = {"img_size": np.array([[1242, 375], [1224, 370], [1408, 376], [1920, 1280]]), "height_crop": np.array([1.0, 1.0, 1.1, 0.9]),
            "canonical_scale": np.array([0.7, 0.9, 0.36, 1.2]), "img_id": np.arange(B) + 10}
    cals = [PinholeCalib(np.array([[700. + 10 * i, 0, 600., 40.], [0, 705., 180., 2.], [0, 0, 1, 0.003]])) for i in range(B)]
    mean = rng.uniform(0, 1, (3, 3)).astype(np.float32)
    a = decode_detections(dets.copy(), info, cals, mean, threshold=0.2)
    b = decode_detections_loop(dets.copy(), info, cals, mean, threshold=0.2)
    assert list(a) == list(b)
    for k in a:
        assert len(a[k]) == len(b[k])
        for ra, rb in zip(a[k], b[k]):
            assert ra[0] == rb[0]
            np.testing.assert_allclose(np.asarray(ra[1:], np.float64), np.asarray(rb[1:], np.float64), rtol=1e-6, atol=1e-6)


def test_all_valid_shortcut_equals_masked_path(monkeypatch):
    """MonoDETR's backbone builds all-False masks; the transformer then skips the masking passes.  Same outputs as
    the general path fed with explicit all-False masks."""
    import monosowa_amd.ms_deform_attn_func as F
    from oracle import msda_oracle as O
    from monosowa_amd.monodetr.depthaware_transformer import DepthAwareTransformer, MLP

    class _Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, w, step):
            return O.msda_core_torch(value, shapes, loc, w)
    monkeypatch.setattr(F, "MSDeformAttnFunction", _Fn)
    torch.manual_seed(0)
    t = DepthAwareTransformer(d_model=256, nhead=8, num_encoder_layers=1, num_decoder_layers=2, dim_feedforward=64, dropout=0.0,
                              return_intermediate_dec=True, group_num=2).eval()
    t.decoder.bbox_embed = torch.nn.ModuleList([MLP(256, 256, 6, 3) for _ in range(2)])
    t.decoder.dim_embed = torch.nn.ModuleList([MLP(256, 256, 3, 2) for _ in range(2)])
    sizes = [(6, 8), (3, 4), (2, 2), (1, 1)]
    srcs = [torch.randn(2, 256, h, w) for h, w in sizes]
    masks = [torch.zeros(2, h, w, dtype=torch.bool) for h, w in sizes]
    pos = [torch.randn(2, 256, h, w) for h, w in sizes]
    qe, dpe = torch.randn(50, 512), torch.randn(2, 256, 3, 4)
    with torch.no_grad():
        a = t(srcs, masks, pos, qe, dpe, dpe, all_valid=False)
        b = t(srcs, masks, pos, qe, dpe, dpe, all_valid=True)
    for x, y in zip(a[:4], b[:4]):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)


def test_block_cost_pass_is_bitwise_the_gathered_cross_matrix():
    """HungarianMatcher.cost_blocks (per-image blocks only) must produce exactly the floats of the reference's full
    cross matrix at those positions: the assignments are index bookkeeping."""
    from monosowa_amd.monodetr import matcher as M
    cfg = _cfg()["model"]
    m = M.build_matcher(cfg)
    g = torch.Generator().manual_seed(4)
    NL, B, Q = 3, 4, 44
    logits = torch.randn(NL, B, Q, 3, generator=g)
    boxes = torch.rand(NL, B, Q, 6, generator=g) * 0.4 + 0.1
    sizes = [3, 0, 5, 2]
    T = sum(sizes)
    flat = {"labels": torch.randint(0, 3, (T,), generator=g), "boxes_3d": torch.rand(T, 6, generator=g) * 0.4 + 0.1}
    res = {}
    for flag in (False, True):
        M.BLOCK_COST = flag
        try:
            (blocks, _), *_ = m.match_layers_begin(logits, boxes, flat, sizes, 11)
        finally:
            M.BLOCK_COST = True
        res[flag] = blocks.clone()
    for b, n in enumerate(sizes):                                   # padding columns are never read by the solver
        assert torch.equal(res[True][:, b, :, :n], res[False][:, b, :, :n])
    a = m.match_layers(logits, boxes, flat, sizes, 11)
    M.BLOCK_COST = False
    try:
        ref = m.match_layers(logits, boxes, flat, sizes, 11)
    finally:
        M.BLOCK_COST = True
    for la, lr in zip(a, ref):
        for (s1, t1), (s2, t2) in zip(la, lr):
            assert np.array_equal(s1, s2) and np.array_equal(t1, t2)


def test_folded_weight_cache_is_invalidated_by_optimizer_steps():
    """The frozen-BN-folded convolution weights are cached when no gradient flows (inference, frozen layers); an
    optimizer step that rewrites the parameters (even behind autograd's back) must invalidate the cache."""
    from monosowa_amd.monodetr import backbone as BB
    conv = torch.nn.Conv2d(4, 8, 1, bias=False)
    bn = BB.FrozenBatchNorm2d(8)
    bn.weight.uniform_(0.5, 1.5)
    x = torch.randn(1, 4, 3, 3)
    with torch.no_grad():
        y0 = BB.conv_bn(x, conv, bn, relu=False).clone()
        w_cached = BB.folded_weight(conv, bn, bn.scale_shift()[0])
        assert BB.folded_weight(conv, bn, bn.scale_shift()[0]) is w_cached            # reused
    opt = torch.optim.SGD(conv.parameters(), lr=0.5)
    conv(x).sum().backward()
    conv.weight.grad.fill_(1.0)
    opt.step()
    with torch.no_grad():
        y1 = BB.conv_bn(x, conv, bn, relu=False)
        ref = bn(conv(x))
    assert not torch.allclose(y0, y1) and torch.allclose(y1, ref, atol=1e-6)


def test_resnet_body_matches_the_published_architecture():
    """Row a10: torchvision is absent, so the backbone cannot be pinned to reference OUTPUTS.  What can be pinned are the
    published facts of torchvision's resnet50 / resnet101 (the models backbone.py:106-108 instantiates): parameter counts
    without the fc layer (25,557,032 - 2,049,000 and 44,549,160 - 2,049,000), torchvision's state-dict key names (a released
    checkpoint loads by name), the v1.5 stride placement (stride 2 in the 3x3 convolution of a stage's first block) and the
    frozen-BN buffers."""
    from monosowa_amd.monodetr.backbone import ResNetBody
    for name, want, blocks in (("resnet50", 23508032, (3, 4, 6, 3)), ("resnet101", 42500160, (3, 4, 23, 3))):
        m = ResNetBody(name)
        sd = m.state_dict()
        learnable = sum(v.numel() for k, v in sd.items() if not k.endswith(("running_mean", "running_var")))
        assert learnable == want, (name, learnable)
        assert sd["conv1.weight"].shape == (64, 3, 7, 7)
        for stage, n in zip((1, 2, 3, 4), blocks):
            layer = getattr(m, "layer%d" % stage)
            assert len(layer) == n
            for i, blk in enumerate(layer):
                prefix = "layer%d.%d." % (stage, i)
                for k in ("conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var", "conv2.weight", "conv3.weight", "bn3.weight"):
                    assert prefix + k in sd, prefix + k
                want_stride = 2 if (i == 0 and stage > 1) else 1
                assert blk.conv1.stride == (1, 1) and blk.conv2.stride == (want_stride, want_stride) and blk.conv2.kernel_size == (3, 3)
                assert (prefix + "downsample.0.weight" in sd) == (i == 0)
        width = 64 * 2 ** 3 * 4
        assert sd["layer4.%d.conv3.weight" % (blocks[3] - 1)].shape[0] == width == 2048
        assert not any(k.endswith("num_batches_tracked") for k in sd)       # FrozenBatchNorm2d drops it (backbone.py:44-52)


def test_dataloader_fed_batches_keep_the_loop_contract():
    """bench.py's DataLoader-fed leg and Trainer.train_one_epoch share ``build_dataloader`` + ``stage_batch`` (reference:
    lib/helpers/dataloader_helper.py:21-34, trainer_helper.py:121-127): a collated batch keeps the (inputs, calibs, targets, info)
    contract, drop_last / test=False behave, and the object mask stays on the host next to its staged copy so that
    ``prepare_targets`` needs no device synchronisation."""
    from monosowa_amd.helpers.dataloader_helper import build_dataloader
    from monosowa_amd.helpers.trainer_helper import stage_batch
    from monosowa_amd.synthetic import prepare_targets
    cfg = {"type": "synthetic", "batch_size": 3, "train_split": "train", "test_split": "val", "resolution": (64, 32), "num_samples": 8}
    loader, test_loader = build_dataloader(cfg, workers=0, drop_last=True, test=False)
    assert test_loader is None and len(loader) == 2                       # 8 samples, batches of 3, the ragged last one dropped
    seen = 0
    for raw in loader:
        inputs, calibs, targets, info = stage_batch(raw, torch.device("cpu"))
        assert tuple(inputs.shape) == (3, 3, 32, 64) and inputs.dtype == torch.float32
        assert tuple(calibs.shape) == (3, 3, 4) and tuple(targets["boxes_3d"].shape) == (3, 50, 6)
        assert targets["labels"].dtype == torch.int8 and targets["heading_bin"].dtype == torch.int64 and targets["mask_2d"].dtype == torch.bool
        host = getattr(targets["mask_2d"], "_host_mask", None)
        assert host is not None and (host == targets["mask_2d"].numpy()).all()
        tl = prepare_targets(targets, 3)
        assert [t["boxes"].shape[0] for t in tl] == host.sum(1).tolist()
        seen += 1
    assert seen == 2
    both = build_dataloader(cfg, workers=0)
    assert both[1] is not None and len(both[0]) == 3                      # the reference's behaviour: nothing dropped, a test loader


def test_dense_flop_bookkeeping_of_the_own_kernels():
    """monosowa_amd/flops.py: off by default (nothing recorded), counts 2 flop per multiply-add for the hand-written dense kernels when
    bench.py's roofline.step switches it on; the attention backward counts the recomputed S once."""
    from monosowa_amd import flops
    flops.attention_forward(2, 8, 100, 200)            # not counting: no effect, no error
    assert flops.stop() == {}
    flops.start()
    flops.attention_forward(2, 8, 100, 200)
    flops.attention_backward(2, 8, 100, 200)
    flops.conv1x1(1000, 64, 256)
    flops.conv1x1(1000, 64, 256)
    flops.linear_wgrad(8800, 256, 128)
    got = flops.stop()
    assert got == {"attention_fwd": 2 * 2 * 2 * 8 * 100 * 200 * 32, "attention_bwd": 5 * 2 * 2 * 8 * 100 * 200 * 32,
                   "conv1x1_fused": 2 * 2 * 1000 * 64 * 256, "linear_wgrad": 2 * 8800 * 256 * 128}
    assert flops.stop() == {}                           # stopped: a second stop has nothing


def test_build_source_hash_follows_the_sources(tmp_path, monkeypatch):
    """monosowa_amd/build.py: a library is rebuilt when the content hash of csrc/ + include/ + flags differs from the one recorded beside
    it -- the hash must change with a source byte and with a flag, and not with file times."""
    import os
    import shutil
    from monosowa_amd import build
    root = tmp_path / "pkg"
    (root / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (root / "csrc" / "a.hip").write_text("int a;\n")
    (tmp_path / "include" / "a.h").write_text("int a();\n")
    monkeypatch.setattr(build, "HERE", str(root))
    monkeypatch.setattr(build, "CSRC", str(root / "csrc"))
    h0 = build.source_hash(["-O3"])
    os.utime(root / "csrc" / "a.hip", (1, 1))
    assert build.source_hash(["-O3"]) == h0
    assert build.source_hash(["-O2"]) != h0
    (root / "csrc" / "a.hip").write_text("int a ;\n")
    assert build.source_hash(["-O3"]) != h0
    (root / "csrc" / "a.hip").write_text("int a;\n")
    assert build.source_hash(["-O3"]) == h0
    (tmp_path / "include" / "a.h").write_text("int a(void);\n")
    assert build.source_hash(["-O3"]) != h0
    # _build_one: recompiles when the recorded hash differs, says "up to date" when it matches
    out = str(tmp_path / "lib.so")
    calls = []
    monkeypatch.setattr(build.subprocess, "check_call", lambda cmd: (calls.append(cmd), open(out, "w").write("x"))[0])
    assert build._build_one(out, ["cc"], "h1", False, False) is True
    assert build._build_one(out, ["cc"], "h1", False, False) is False
    assert build._build_one(out, ["cc"], "h2", False, False) is True
    assert build._build_one(out, ["cc"], "h2", True, False) is True
    assert len(calls) == 3


def test_pooled_zero_accumulators_are_zero_disjoint_and_aligned(monkeypatch):
    """pointwise.zeros_f64: the GroupNorm kernels' zeroed float64 accumulators come as slices of one pooled chunk (one fill per chunk,
    not one per accumulator); a slice is handed out once, a request that would not fit starts a new chunk, a large one bypasses the pool."""
    import torch
    from monosowa_amd import pointwise as pw
    monkeypatch.setattr(pw, "ZERO_POOL_DOUBLES", 64)
    monkeypatch.setattr(pw, "_ZERO_POOL", {})
    dev = torch.device("cpu")
    a = pw.zeros_f64(5, dev); b = pw.zeros_f64(6, dev); c = pw.zeros_f64(16, dev)
    assert a.dtype == torch.float64 and a.shape == (5,) and b.shape == (6,) and c.shape == (16,)
    assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and c.data_ptr() % 16 == 0
    assert b.data_ptr() >= a.data_ptr() + 5 * 8 and c.data_ptr() >= b.data_ptr() + 6 * 8          # disjoint, same chunk
    a += 1.0; b += 2.0
    assert float(c.abs().sum()) == 0.0 and float(a.sum()) == 5.0 and float(b.sum()) == 12.0
    chunk0 = pw._ZERO_POOL[("cpu", None)][0]
    for _ in range(3):
        d = pw.zeros_f64(16, dev)                                        # 6 + 6 + 16 + 16 + 16 + 16 > 64: a new chunk on the way
        assert float(d.abs().sum()) == 0.0
        d += 3.0
    assert pw._ZERO_POOL[("cpu", None)][0] is not chunk0
    big = pw.zeros_f64(17, dev)                                          # more than a quarter of a chunk: its own allocation
    assert big.shape == (17,) and float(big.abs().sum()) == 0.0
    assert float(a.sum()) == 5.0                                         # earlier slices untouched
