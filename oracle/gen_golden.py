"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REFERENCE's own Python.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [msda] [module] [transformer] [depth] [misc]
                                                       [adamw] [decode] [criterion] [heads] [kitti_eval] [kitti_dataset] [kitti_dataset_pd]

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


def _fill():
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from det_weights import fill_deterministic, key_manifest
    return fill_deterministic, key_manifest


KITTI_SMALL = [(12, 16), (6, 8), (3, 4), (2, 2)]     # a 128x96 image at strides 8,16,32,64


def gen_module():
    """MSDeformAttn module (ops/modules/ms_deform_attn.py:69-162): 2-d and 6-d reference points."""
    imp = ref_pkg()
    fill, manifest = _fill()
    mod = imp.import_module("mdpkg.ops.modules.ms_deform_attn")
    torch.manual_seed(21)
    m = fill(mod.MSDeformAttn(256, 4, 8, 4), 101).double()
    shapes = torch.as_tensor(KITTI_SMALL, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    B, Lq = 2, 37
    query = torch.randn(B, Lq, 256, dtype=torch.double)
    src = torch.randn(B, S, 256, dtype=torch.double)
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[1, -5:] = True
    ref2 = torch.rand(B, Lq, 4, 2, dtype=torch.double)
    ref6 = torch.cat([torch.rand(B, Lq, 4, 2, dtype=torch.double), torch.rand(B, Lq, 4, 4, dtype=torch.double) * 0.2], -1)
    out2 = m(query, ref2, src, shapes, _lsi(shapes), pad)
    out6 = m(query, ref6, src, shapes, _lsi(shapes), pad)
    _save("module_msdeformattn", query=query, src=src, pad=pad, ref2=ref2, ref6=ref6, shapes=shapes,
          out2=out2, out6=out6, manifest=manifest(m))


def _transformer(imp, group_num, dropout=0.0):
    dt = imp.import_module("mdpkg.depthaware_transformer")
    t = dt.DepthAwareTransformer(d_model=256, nhead=8, num_encoder_layers=3, num_decoder_layers=3,
                                 dim_feedforward=256, dropout=dropout, return_intermediate_dec=True,
                                 num_feature_levels=4, dec_n_points=4, enc_n_points=4, two_stage=False,
                                 two_stage_num_proposals=50, group_num=group_num)
    # what MonoDETR.__init__ attaches (monodetr.py:130-137)
    t.decoder.bbox_embed = torch.nn.ModuleList([dt.MLP(256, 256, 6, 3) for _ in range(3)])
    t.decoder.dim_embed = torch.nn.ModuleList([dt.MLP(256, 256, 3, 2) for _ in range(3)])
    return t


def gen_transformer():
    """Whole DepthAwareTransformer (3 enc + 3 dec, iterative refinement) at reduced resolution, eval
    (50 queries) and train (3 groups x 50, dropout 0); one encoder layer and one decoder layer alone."""
    imp = ref_pkg()
    fill, manifest = _fill()
    torch.manual_seed(31)
    G = 3
    t = fill(_transformer(imp, G), 202).double()      # reference run in float64; stored rounded to f32
    B = 2
    srcs = [torch.randn(B, 256, h, w).double() for h, w in KITTI_SMALL]
    masks = [torch.zeros(B, h, w, dtype=torch.bool) for h, w in KITTI_SMALL]
    masks[0][1, :, -3:] = True     # some padding on the right of image 1 (all levels consistently)
    masks[1][1, :, -2:] = True
    masks[2][1, :, -1:] = True
    pos = [(torch.randn(B, 256, h, w) * 0.5).double() for h, w in KITTI_SMALL]
    query_embed = torch.randn(G * 50, 512).double()
    depth_pos_embed = torch.randn(B, 256, 6, 8).double()
    out = {}
    for mode in ("eval", "train"):
        t.train(mode == "train")
        q = query_embed if mode == "train" else query_embed[:50]
        hs, init_ref, inter_refs, inter_dims, _, _ = t(srcs, masks, pos, q, depth_pos_embed, depth_pos_embed)
        out.update({mode + "_hs": hs, mode + "_init_ref": init_ref, mode + "_inter_refs": inter_refs,
                    mode + "_inter_dims": inter_dims})
    arrays = {"src%d" % i: s for i, s in enumerate(srcs)}
    arrays.update({"mask%d" % i: m for i, m in enumerate(masks)})
    arrays.update({"pos%d" % i: p for i, p in enumerate(pos)})
    # gradient of a scalar through the whole train-mode transformer w.r.t. src0 (autograd through MSDA)
    t.train(True)
    s0 = srcs[0].clone().requires_grad_(True)
    hs = t([s0] + srcs[1:], masks, pos, query_embed, depth_pos_embed, depth_pos_embed)[0]
    (hs * torch.linspace(-1, 1, hs.numel(), dtype=torch.double).view_as(hs)).sum().backward()
    f32 = lambda d: {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.double else v) for k, v in d.items()}
    _save("transformer_small", query_embed=query_embed.float(), depth_pos_embed=depth_pos_embed.float(), group_num=G,
          grad_src0=s0.grad.float(), manifest=manifest(t), **f32(arrays), **f32(out))


def gen_depth():
    """DepthPredictor (depth_predictor/depth_predictor.py:56-104) incl. the integer floor indices."""
    _shims()
    if MD not in sys.path:
        sys.path.insert(0, MD)
    import importlib
    dp = importlib.import_module("depth_predictor.depth_predictor")
    fill, manifest = _fill()
    cfg = {"num_depth_bins": 80, "depth_min": 1e-3, "depth_max": 60.0, "hidden_dim": 256}
    torch.manual_seed(41)
    m = fill(dp.DepthPredictor(cfg), 303).eval()
    B = 2
    feats = [torch.randn(B, 256, h, w) for h, w in KITTI_SMALL]
    mask = torch.zeros(B, 6, 8, dtype=torch.bool)
    mask[1, :, -2:] = True
    pos = torch.randn(B, 256, 6, 8) * 0.5
    logits, embed, wdepth, ip = m(feats, mask, pos)
    floor_idx = wdepth.clamp(min=0, max=60.0).floor().long()
    _save("depth_predictor", f0=feats[0], f1=feats[1], f2=feats[2], f3=feats[3], mask=mask, pos=pos,
          logits=logits, embed=embed, weighted_depth=wdepth, ip=ip, floor_idx=floor_idx,
          bin_values=m.depth_bin_values, manifest=manifest(m))


def gen_misc():
    """HungarianMatcher indices, sine position encoding, focal losses, DDN depth-map loss pieces."""
    imp = ref_pkg()
    torch.manual_seed(51)
    matcher = imp.import_module("mdpkg.matcher")
    B, Q, G = 3, 100, 2
    outputs = {"pred_logits": torch.randn(B, Q, 3), "pred_boxes": torch.rand(B, Q, 6) * 0.3 + 0.05}
    outputs["pred_boxes"][..., :2] += 0.3
    sizes = [4, 1, 7]
    targets, flat = [], {}
    for i, n in enumerate(sizes):
        b3 = torch.cat([torch.rand(n, 2) * 0.8 + 0.1, torch.rand(n, 4) * 0.09 + 0.01], 1)
        t = {"labels": torch.randint(0, 3, (n,)).to(torch.int8), "boxes_3d": b3, "boxes": torch.rand(n, 4)}
        targets.append(t)
        for k, v in t.items():
            flat["t%d_%s" % (i, k)] = v
    m = matcher.HungarianMatcher(cost_class=2, cost_3dcenter=10, cost_bbox=5, cost_giou=2)
    ind = m(outputs, targets, group_num=G)
    for i, (a, b) in enumerate(ind):
        flat["ind%d_src" % i], flat["ind%d_tgt" % i] = a, b
    _save("matcher", pred_logits=outputs["pred_logits"], pred_boxes=outputs["pred_boxes"], group_num=G,
          sizes=np.array(sizes), **flat)

    pe = imp.import_module("mdpkg.position_encoding")
    misc = importlib_misc()
    mask = torch.zeros(2, 6, 8, dtype=torch.bool)
    mask[1, :, -3:] = True
    mask[1, -1:, :] = True
    pos = pe.PositionEmbeddingSine(128, normalize=True)(misc.NestedTensor(torch.zeros(2, 256, 6, 8), mask))
    _save("position_sine", mask=mask, pos=pos)

    # losses: sigmoid focal (lib/losses/focal_loss.py:69-94), DDN focal + balancer + LID binning
    sys.path.insert(0, REF)
    import importlib
    fl = importlib.import_module("lib.losses.focal_loss")
    logits = torch.randn(2, 30, 3)
    tgt = (torch.rand(2, 30, 3) > 0.8).float()
    sfl = fl.sigmoid_focal_loss(logits, tgt, 7.0, alpha=0.25, gamma=2)
    ddn = importlib.import_module("mdpkg.depth_predictor.ddn_loss.ddn_loss")
    saved = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0          # DDNLoss.__init__ only stores it (ddn_loss.py:32)
    try:
        crit = ddn.DDNLoss()
    finally:
        torch.cuda.current_device = saved
    depth_logits = torch.randn(2, 81, 24, 80)
    num_gt = [3, 2]
    boxes = torch.tensor([[10.2, 3.7, 30.9, 12.1], [-2.5, 5.0, 8.3, 20.0], [50.0, 0.2, 79.7, 23.9],
                          [20.5, 8.5, 26.1, 13.3], [24.0, 10.0, 40.0, 18.0]])
    depth = torch.tensor([12.5, 40.0, 7.25, 61.0, 30.0])
    ddn_loss = crit(depth_logits, boxes.clone(), num_gt, depth)
    dm = crit.build_target_depth_from_3dcenter(depth_logits, boxes.clone(), depth, num_gt)
    bins = crit.bin_depths(dm, target=True)
    _save("losses", sfl_logits=logits, sfl_targets=tgt, sfl=sfl, depth_logits=depth_logits, boxes=boxes,
          depth=depth, num_gt=np.array(num_gt), ddn_loss=ddn_loss, depth_map=dm, depth_bins=bins)


# ----------------------------------------------------------------------------- round 2: AdamW, decode, criterion, heads
def _more_shims():
    """Absent third-party modules that reference files import at module level but never call on the paths
    fixtured here: open3d / cv2 (visualisation, image IO) and torchvision.models (the ResNet body, replaced
    by a feature stub below)."""
    _shims()
    for name in ("open3d", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "models"):
        models = types.ModuleType("torchvision.models")
        utils = types.ModuleType("torchvision.models._utils")

        class IntermediateLayerGetter(torch.nn.Module):       # only named at import time (backbone.py:19)
            pass
        utils.IntermediateLayerGetter = IntermediateLayerGetter
        models._utils = utils
        tv.models = models
        sys.modules.update({"torchvision.models": models, "torchvision.models._utils": utils})


class _NoCuda:
    """The reference hard-codes ``.cuda()`` / ``device='cuda'`` in its loss methods (monodetr.py:515,528,567-575,
    ddn_loss.py:32).  Inside this context those land on the CPU: torch is patched, the reference is not."""

    def __enter__(self):
        self._saved = (torch.Tensor.cuda, torch.tensor, torch.cuda.current_device)
        orig_tensor = torch.tensor

        def tensor(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k.pop("device")
            return orig_tensor(*a, **k)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.tensor = tensor
        torch.cuda.current_device = lambda: 0
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.tensor, torch.cuda.current_device = self._saved


def gen_adamw():
    """The reference's own AdamW (lib/helpers/optimizer_helper.py:30-129): three steps on two parameter groups
    (weight decay 0 / 1e-4) with tensors whose sizes exercise the fused kernel's aligned and unaligned paths."""
    _shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    import warnings
    oh = importlib.import_module("lib.helpers.optimizer_helper")
    g = torch.Generator().manual_seed(71)
    shapes = [(7,), (33,), (256,), (5, 3), (48, 64), (4097,), (3, 3, 3, 16), (1,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    arrays = {"p%d_init" % i: p.detach().clone() for i, p in enumerate(params)}
    groups = [{"params": params[0::2], "weight_decay": 0.0}, {"params": params[1::2], "weight_decay": 1e-4}]
    opt = oh.AdamW(groups, lr=2e-4)
    n_steps = 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(n_steps):
            for i, p in enumerate(params):
                grad = torch.randn(p.shape, generator=g) * (10.0 ** ((i % 3) - 1))
                if step == 1 and i == 2:
                    grad = torch.zeros_like(grad)          # a zero gradient: sqrt(v) + eps path
                p.grad = grad
                arrays["g%d_step%d" % (i, step)] = grad.clone()
            opt.step()
            for i, p in enumerate(params):
                arrays["p%d_step%d" % (i, step)] = p.detach().clone()
    for i, p in enumerate(params):
        arrays["m%d" % i] = opt.state[p]["exp_avg"].clone()
        arrays["v%d" % i] = opt.state[p]["exp_avg_sq"].clone()
    _save("adamw", n_params=len(params), n_steps=n_steps, lr=2e-4, weight_decay=1e-4, **arrays)


def _seeded_outputs(gen, B, Q, dtype=torch.float32):
    o = {"pred_logits": torch.randn(B, Q, 3, generator=gen) * 2 - 1.5,
         "pred_boxes": torch.cat([torch.rand(B, Q, 2, generator=gen) * 0.8 + 0.1,
                                  torch.rand(B, Q, 4, generator=gen) * 0.15 + 0.01], -1),
         "pred_3d_dim": torch.randn(B, Q, 3, generator=gen) * 0.2 + torch.tensor([1.5, 1.6, 3.9]),
         "pred_depth": torch.cat([torch.rand(B, Q, 1, generator=gen) * 55 + 3, torch.randn(B, Q, 1, generator=gen) * 0.5], -1),
         "pred_angle": torch.randn(B, Q, 24, generator=gen)}
    return {k: v.to(dtype) for k, v in o.items()}


def gen_decode():
    """extract_dets_from_outputs + decode_detections (lib/helpers/decode_helper.py:8-111) with the reference's own
    Calibration class (lib/datasets/kitti/kitti_utils.py:137-284) built from P2 matrices."""
    _more_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    dh = importlib.import_module("lib.helpers.decode_helper")
    ku = importlib.import_module("lib.datasets.kitti.kitti_utils")
    gen = torch.Generator().manual_seed(81)
    B, Q = 3, 50
    outputs = _seeded_outputs(gen, B, Q)
    outputs["pred_logits"][0, 3] = torch.tensor([3.0, 2.5, 2.0])         # exact score ties well inside the top-k (their
    outputs["pred_logits"][0, 7] = outputs["pred_logits"][0, 3]          # relative order is device-dependent, membership is not)
    outputs["pred_logits"][1, :, 1] = 4.0 - torch.arange(Q) * 0.01       # many confident detections
    dets = dh.extract_dets_from_outputs(outputs, K=50, topk=50)
    P2 = np.array([[[721.54, 0, 609.56, 44.857], [0, 721.54, 172.85, 0.2163], [0, 0, 1, 0.002746]],
                   [[552.55, 0, 682.05, -328.3], [0, 552.55, 238.77, 0.0], [0, 0, 1, 0.0]],
                   [[2055.6, 0, 939.66, 0.0], [0, 2055.6, 641.07, 0.0], [0, 0, 1, 0.0]]], dtype=np.float32)
    calibs = [ku.Calibration({"P2": P2[i], "R0": np.eye(3, dtype=np.float32),
                              "Tr_velo2cam": np.eye(3, 4, dtype=np.float32)}) for i in range(B)]
    info = {"img_id": np.array([11, 22, 33]), "img_size": np.array([[1242, 375], [1408, 376], [1920, 1280]]),
            "height_crop": np.array([1.0, 1.0, 2.5]), "canonical_scale": np.array([0.693, 0.905, 0.243])}
    cls_mean_size = np.array([[1.76255119, 0.66068622, 0.84422524], [1.52563191, 1.62856739, 3.52588311],
                              [1.73698127, 0.59706367, 1.76282397]])
    res = dh.decode_detections(dets.detach().numpy().copy(), info, calibs, cls_mean_size, 0.2)
    rows = np.zeros((B, 50, 14), dtype=np.float64)
    counts = np.zeros(B, dtype=np.int64)
    for i, img in enumerate(info["img_id"]):
        counts[i] = len(res[img])
        for j, r in enumerate(res[img]):
            rows[i, j] = np.asarray(r, dtype=np.float64)
    _save("decode", P2=P2, dets=dets, decoded=rows, counts=counts, cls_mean_size=cls_mean_size, threshold=0.2,
          **{"info_" + k: v for k, v in info.items()}, **outputs)


def _seeded_targets(gen, sizes):
    targets = []
    for n in sizes:
        c = torch.rand(n, 2, generator=gen) * 0.8 + 0.1
        lrtb = torch.rand(n, 4, generator=gen) * 0.09 + 0.01
        b3 = torch.cat([c, lrtb], 1)
        cx = c[:, 0] + (lrtb[:, 1] - lrtb[:, 0]) / 2
        cy = c[:, 1] + (lrtb[:, 3] - lrtb[:, 2]) / 2
        boxes = torch.stack([cx, cy, lrtb[:, 0] + lrtb[:, 1], lrtb[:, 2] + lrtb[:, 3]], 1)
        targets.append({"labels": torch.randint(0, 3, (n,), generator=gen).to(torch.int8), "boxes": boxes, "boxes_3d": b3,
                        "depth": torch.rand(n, 1, generator=gen) * 55 + 5,
                        "size_3d": torch.randn(n, 3, generator=gen) * 0.1 + torch.tensor([1.53, 1.63, 3.88]),
                        "heading_bin": torch.randint(0, 12, (n, 1), generator=gen),
                        "heading_res": (torch.rand(n, 1, generator=gen) - 0.5) * (np.pi / 6)})
    return targets


CRIT_WEIGHTS = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0, "loss_dim": 1.0, "loss_angle": 1.0, "loss_depth": 1.0,
                "loss_center": 10.0, "loss_depth_map": 1.0, "loss_tfl": 0.0, "loss_mask": 0.0}


def gen_criterion():
    """SetCriterion.forward (monodetr.py:1188-1230) with its own loss methods (:396-536) and the reference matcher, train
    mode (3 groups x 50 queries, two auxiliary layers): every entry of the loss dict plus the gradient of the weighted
    total (trainer_helper.py:140-141) w.r.t. every prediction tensor."""
    _more_shims()
    imp = ref_pkg()
    mono = imp.import_module("mdpkg.monodetr")
    matcher = imp.import_module("mdpkg.matcher")
    gen = torch.Generator().manual_seed(91)
    B, G, NL = 3, 3, 3
    Q = 50 * G
    sizes = [4, 1, 7]
    targets = _seeded_targets(gen, sizes)
    layers = [_seeded_outputs(gen, B, Q) for _ in range(NL)]
    depth_map_logits = torch.randn(B, 81, 24, 80, generator=gen)
    leaves = []
    for o in layers:
        for k in o:
            o[k].requires_grad_(True)
            leaves.append(o[k])
    depth_map_logits.requires_grad_(True)
    outputs = dict(layers[-1])
    outputs["pred_depth_map_logits"] = depth_map_logits
    outputs["aux_outputs"] = layers[:-1]
    weight_dict = dict(CRIT_WEIGHTS)
    for i in range(NL - 1):
        weight_dict.update({k + "_%d" % i: v for k, v in CRIT_WEIGHTS.items()})
    losses = ["labels", "boxes", "cardinality", "depths", "dims", "angles", "center", "depth_map", "tfl"]
    with _NoCuda():
        m = matcher.HungarianMatcher(cost_class=2, cost_3dcenter=10, cost_bbox=5, cost_giou=2)
        crit = mono.SetCriterion(3, matcher=m, weight_dict=weight_dict, focal_alpha=0.25, losses=losses, group_num=G)
        crit.train()
        loss_dict = crit(outputs, targets)
        total = sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict)
        total.backward()
    arrays = {"sizes": np.array(sizes), "group_num": G, "n_layers": NL, "total": total.detach(),
              "depth_map_logits": depth_map_logits.detach(), "grad_depth_map_logits": depth_map_logits.grad}
    for li, o in enumerate(layers):                      # layer NL-1 = final outputs, 0 .. NL-2 = aux_outputs
        for k, v in o.items():
            arrays["l%d_%s" % (li, k)] = v.detach()
            arrays["l%d_grad_%s" % (li, k)] = v.grad if v.grad is not None else torch.zeros_like(v)
    for i, t in enumerate(targets):
        for k, v in t.items():
            arrays["t%d_%s" % (i, k)] = v
    for k, v in loss_dict.items():
        arrays["loss__" + k] = v.detach() if torch.is_tensor(v) else torch.tensor(float(v))
    arrays["weight_keys"] = np.array(sorted(weight_dict))
    arrays["weight_vals"] = np.array([weight_dict[k] for k in sorted(weight_dict)], dtype=np.float64)
    _save("criterion", **arrays)


class _FeatureStub(torch.nn.Module):
    """Stands in for torchvision's ResNet body (absent here): returns fixed C3/C4/C5 maps with all-False masks, as
    BackboneBase.forward does (backbone.py:85-91).  Everything downstream is the reference's own code."""
    strides = [8, 16, 32]
    num_channels = [512, 1024, 2048]

    def __init__(self, feats, NestedTensor):
        super().__init__()
        self.feats, self.NT = feats, NestedTensor

    def forward(self, images):
        return {str(i): self.NT(f, torch.zeros(f.shape[0], f.shape[2], f.shape[3], dtype=torch.bool))
                for i, f in enumerate(self.feats)}


def gen_heads():
    """MonoDETR.forward behind the backbone body (monodetr.py:155-289): the reference's Joiner + sine encoding, input_proj,
    extra level, DepthPredictor, transformer and all per-layer heads, eval (50 queries) and train (3 x 50, dropout off)."""
    _more_shims()
    imp = ref_pkg()
    fill, manifest = _fill()
    mono = imp.import_module("mdpkg.monodetr")
    bb = imp.import_module("mdpkg.backbone")
    pe = imp.import_module("mdpkg.position_encoding")
    dp = imp.import_module("mdpkg.depth_predictor")
    misc = importlib_misc()
    torch.manual_seed(61)
    G, B = 3, 2
    feats = [torch.randn(B, c, h, w).double() * 0.5 for c, (h, w) in zip((512, 1024, 2048), KITTI_SMALL[:3])]
    backbone = bb.Joiner(_FeatureStub(feats, misc.NestedTensor), pe.PositionEmbeddingSine(128, normalize=True))
    cfg = {"num_depth_bins": 80, "depth_min": 1e-3, "depth_max": 60.0, "hidden_dim": 256}
    model = mono.MonoDETR(backbone, _transformer(imp, G, dropout=0.0), dp.DepthPredictor(cfg), num_classes=3, num_queries=50,
                          num_feature_levels=4, aux_loss=True, with_box_refine=True, two_stage=False, init_box=False,
                          use_dab=False, group_num=G, two_stage_dino=False)
    for mod in model.modules():                          # the depth encoder ships dropout 0.1: switch the randomness off
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    model = fill(model, 404).double()
    images = torch.zeros(B, 3, 96, 128).double()
    calibs = torch.tensor([[[707.05, 0, 640.0, 0], [0, 707.05, 192.0, 0], [0, 0, 1, 0]],
                           [[552.6, 0, 682.0, 0], [0, 552.6, 238.8, 0], [0, 0, 1, 0]]]).double()
    img_sizes = torch.tensor([[1242.0, 375.0], [1408.0, 376.0]]).double()
    arrays = {"f%d" % i: f.float() for i, f in enumerate(feats)}
    for mode in ("eval", "train"):
        model.train(mode == "train")
        out = model(images, calibs, None, img_sizes)
        for k in ("pred_logits", "pred_boxes", "pred_3d_dim", "pred_depth", "pred_angle", "pred_depth_map_logits"):
            arrays["%s_%s" % (mode, k)] = out[k].float()
        for i, aux in enumerate(out["aux_outputs"]):
            for k, v in aux.items():
                arrays["%s_aux%d_%s" % (mode, i, k)] = v.float()
    _save("monodetr_heads", calibs=calibs.float(), img_sizes=img_sizes.float(), group_num=G, manifest=manifest(model), **arrays)



# ----------------------------------------------------------------------------- KITTI evaluation (kitti_eval_python)
def _numba_shim():
    """numba is absent from this image.  The reference's evaluation code is plain Python under its decorators, so an
    identity ``jit`` runs it unchanged (slowly); ``cuda.local.array`` becomes a float32 numpy array, which makes the
    rotated-IoU DEVICE FUNCTIONS (rotate_iou.py:17-259) callable pair by pair in float32 numpy arithmetic.  The CUDA
    kernel / launch code of that file is never called."""
    if "numba" in sys.modules:
        return
    def deco(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb = types.ModuleType("numba")
    nb.jit = deco
    nb.float32 = np.float32
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = deco
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype=np.float32))
    nb.cuda = cuda
    sys.modules.update({"numba": nb, "numba.cuda": cuda})


def ref_kitti_eval():
    _numba_shim()
    import importlib
    if "kepkg" not in sys.modules:
        pkg = types.ModuleType("kepkg")
        pkg.__path__ = [REF + "/lib/datasets/kitti/kitti_eval_python"]
        sys.modules["kepkg"] = pkg
    ev = importlib.import_module("kepkg.eval")
    riou = importlib.import_module("kepkg.rotate_iou")

    def all_pairs(boxes, query_boxes, criterion=-1, device_id=0):
        """rotate_iou_gpu_eval's contract (rotate_iou.py:295-330) by calling the reference DEVICE function per pair."""
        b = np.ascontiguousarray(boxes, dtype=np.float32)
        q = np.ascontiguousarray(query_boxes, dtype=np.float32)
        out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
        for n in range(b.shape[0]):
            for k in range(q.shape[0]):
                out[n, k] = riou.devRotateIoUEval(q[k].copy(), b[n].copy(), criterion)    # kernel argument order: rotate_iou.py:291-293
        return out.astype(boxes.dtype)
    ev.rotate_iou_gpu_eval = all_pairs
    return ev, riou, all_pairs


def _synthetic_annos(rng, n_images):
    names_gt = ["Car", "Car", "Car", "Pedestrian", "Cyclist", "Van", "DontCare", "Person_sitting", "Truck"]
    gts, dts = [], []
    for img in range(n_images):
        n = int(rng.integers(3, 9))
        name = [names_gt[int(rng.integers(0, len(names_gt)))] for _ in range(n)]
        z = rng.uniform(6, 45, n)
        x = rng.uniform(-12, 12, n)
        y = rng.uniform(1.2, 2.0, n)
        dims = np.stack([rng.normal(3.9, 0.4, n), rng.normal(1.55, 0.1, n), rng.normal(1.65, 0.1, n)], 1)   # l, h, w (camera order)
        small = np.array([nm in ("Pedestrian", "Person_sitting", "Cyclist") for nm in name])
        dims[small] = np.stack([rng.normal(0.9, 0.2, small.sum()), rng.normal(1.75, 0.1, small.sum()), rng.normal(0.65, 0.1, small.sum())], 1)
        ry = rng.uniform(-np.pi, np.pi, n)
        hpx = 721.5 * dims[:, 1] / z
        wpx = 721.5 * np.maximum(dims[:, 0], dims[:, 2]) / z * rng.uniform(0.5, 1.0, n)
        cu, cv = 609.5 + 721.5 * x / z, 172.8 + 721.5 * (y - dims[:, 1] / 2) / z
        bbox = np.stack([cu - wpx / 2, cv - hpx / 2, cu + wpx / 2, cv + hpx / 2], 1)
        gt = {"name": np.array(name), "truncated": np.round(rng.choice([0.0, 0.0, 0.0, 0.1, 0.2, 0.4, 0.6], n), 2),
              "occluded": rng.choice([0, 0, 0, 1, 2, 3], n), "alpha": ry - np.arctan2(x, z), "bbox": bbox, "dimensions": dims,
              "location": np.stack([x, y, z], 1), "rotation_y": ry, "score": np.zeros(n)}
        gts.append(gt)
        # detections: noisy copies of most real objects + a few false positives, classes mostly right
        keep = [i for i in range(n) if name[i] != "DontCare" and rng.random() < 0.85]
        m = len(keep)
        fp = int(rng.integers(0, 3))
        d_name = [name[i] if rng.random() < 0.9 else "Car" for i in keep] + ["Car", "Pedestrian", "Cyclist"][:fp]
        noise = lambda s, k: rng.normal(0, s, k)
        d_loc = np.concatenate([gt["location"][keep] + np.stack([noise(0.05, m), noise(0.02, m), noise(0.12, m)], 1) * rng.choice([1.0, 1.0, 4.0], (m, 1)),
                                np.stack([rng.uniform(-10, 10, fp), rng.uniform(1.3, 1.9, fp), rng.uniform(8, 50, fp)], 1)])
        d_dims = np.concatenate([dims[keep] * (1 + 0.02 * rng.standard_normal((m, 3))), np.tile([[3.8, 1.5, 1.6]], (fp, 1))])
        d_ry = np.concatenate([ry[keep] + noise(0.04, m), rng.uniform(-np.pi, np.pi, fp)])
        d_bbox = np.concatenate([bbox[keep] + rng.normal(0, 1.5, (m, 4)),
                                 np.stack([rng.uniform(100, 900, fp), rng.uniform(120, 200, fp), rng.uniform(100, 900, fp) + 60,
                                           rng.uniform(120, 200, fp) + 45], 1).reshape(fp, 4)])
        if fp:
            d_bbox[m:, 2] = d_bbox[m:, 0] + rng.uniform(30, 90, fp)
            d_bbox[m:, 3] = d_bbox[m:, 1] + rng.uniform(20, 70, fp)
        k = m + fp
        dts.append({"name": np.array(d_name) if k else np.zeros(0, dtype="<U3"), "truncated": np.zeros(k), "occluded": np.zeros(k, dtype=np.int64),
                    "alpha": d_ry - np.arctan2(d_loc[:, 0], d_loc[:, 2]), "bbox": d_bbox.reshape(k, 4), "dimensions": d_dims.reshape(k, 3),
                    "location": d_loc.reshape(k, 3), "rotation_y": d_ry, "score": np.round(rng.uniform(0.05, 0.99, k), 4)})
    return gts, dts


def _pack_annos(prefix, annos, out):
    out[prefix + "_count"] = np.array([len(a["name"]) for a in annos])
    for key in ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score"):
        parts = [np.asarray(a[key]) for a in annos if len(a["name"])]
        out[prefix + "_" + key] = np.concatenate(parts) if parts else np.zeros(0)
    out[prefix + "_name"] = out[prefix + "_name"].astype("U16")


def gen_kitti_eval():
    """(1) the reference's rotated-IoU device function on random / degenerate box pairs, all four criteria;
    (2) the whole AP pipeline (clean_data, calculate_iou_partly, compute_statistics_jit, get_thresholds,
    fused_compute_statistics, eval_class, do_eval, get_official_eval_result; eval.py:10-80, :160-260, :234-760) on a seeded
    synthetic set of 24 images, with the overlap matrices the reference used."""
    ev, riou, all_pairs = ref_kitti_eval()
    rng = np.random.default_rng(20241)
    n, k = 36, 44
    boxes = np.stack([rng.uniform(-6, 6, n), rng.uniform(10, 22, n), rng.uniform(1.2, 4.5, n), rng.uniform(1.2, 4.5, n), rng.uniform(-3.2, 3.2, n)], 1)
    qboxes = np.stack([rng.uniform(-6, 6, k), rng.uniform(10, 22, k), rng.uniform(1.2, 4.5, k), rng.uniform(1.2, 4.5, k), rng.uniform(-3.2, 3.2, k)], 1)
    qboxes[:8] = boxes[:8]                                     # identical boxes
    qboxes[8:14, :2] = boxes[8:14, :2] + 0.05                  # nearly coincident centres
    qboxes[14:18, 4] = boxes[14:18, 4] + np.pi / 2             # same centre-ish, rotated a quarter turn
    qboxes[14:18, :2] = boxes[14:18, :2]
    qboxes[18:20] = [[100, 100, 2, 2, 0.3], [-50, 3, 1, 4, 1.0]]        # far away: zero overlap
    boxes, qboxes = boxes.astype(np.float32), qboxes.astype(np.float32)
    out = {"boxes": boxes, "qboxes": qboxes}
    for crit in (-1, 0, 1, 2):
        out["iou_crit%d" % crit] = all_pairs(boxes, qboxes, crit)
    _save("kitti_rotate_iou", **out)

    gts, dts = _synthetic_annos(rng, 24)
    out = {}
    _pack_annos("gt", gts, out)
    _pack_annos("dt", dts, out)
    difficultys = [0, 1, 2]
    min_overlaps = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.5, 0.5, 0.5], [0.5, 0.25, 0.25], [0.5, 0.25, 0.25]])], 0)
    for metric in (0, 1, 2):
        ret = ev.eval_class(gts, dts, [0, 1, 2], difficultys, metric, min_overlaps, compute_aos=(metric == 0))
        out["m%d_precision" % metric], out["m%d_recall" % metric] = ret["precision"], ret["recall"]
        if metric == 0:
            out["m0_orientation"] = ret["orientation"]
        overlaps = ev.calculate_iou_partly(dts, gts, metric, 50)[0]           # per image [n_dt, n_gt], the order eval_class uses
        out["m%d_overlaps" % metric] = np.concatenate([o.reshape(-1) for o in overlaps]) if overlaps else np.zeros(0)
    out["min_overlaps"] = min_overlaps
    lines = []
    for cls in (0, 1, 2):
        result, ret_dict, car_mod = ev.get_official_eval_result(gts, dts, cls)
        lines.append(result)
        for key, val in ret_dict.items():
            out["official_%d__%s" % (cls, key)] = np.float64(val)
        out["official_%d_return" % cls] = np.float64(car_mod)
    out["official_text"] = np.array("\n=====\n".join(lines))
    # the distance-range variant (clean_data_by_distance, get_distance_eval_result: eval.py:83-157, :988-1090)
    lines = []
    for cls in (0, 1):
        result, ret_dict = ev.get_distance_eval_result(gts, dts, cls)
        lines.append(result)
        for key, val in ret_dict.items():
            out["distance_%d__%s" % (cls, key)] = np.float64(val)
    out["distance_text"] = np.array("\n=====\n".join(lines))
    # the pieces, on their own
    sc = np.round(rng.uniform(0, 1, 57), 3)
    out["thr_scores"], out["thr_num_gt"] = sc, np.int64(71)
    out["thr_out"] = np.array(ev.get_thresholds(sc.copy(), 71))
    _save("kitti_ap", **out)


# ----------------------------------------------------------------------------- KITTI file dataset (kitti_dataset.py)
def _dataset_shims():
    """Third-party modules lib/datasets/kitti/kitti_dataset.py imports at module level and that are absent here: cv2, dill,
    zstd, point_cloud_utils, open3d, skimage, torchvision.transforms, numba.  Only ONE of their functions is reached on the
    fixtured path: cv2.getAffineTransform (kitti_utils.py:374-379), shimmed by its definition -- the 2 x 3 matrix mapping
    three points onto three points, solved in float64."""
    _more_shims()
    _numba_shim()
    for name in ("dill", "zstd", "point_cloud_utils", "skimage"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].io = types.ModuleType("skimage.io")
    sys.modules.setdefault("skimage.io", sys.modules["skimage"].io)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision.transforms"] = tv.transforms

    def get_affine_transform(src, dst):
        a = np.hstack([np.asarray(src, dtype=np.float64), np.ones((3, 1))])
        return np.linalg.solve(a, np.asarray(dst, dtype=np.float64)).T
    sys.modules["cv2"].getAffineTransform = get_affine_transform


def _synthetic_kitti_files(rng, ids):
    """-> {relative path: bytes} of a small KITTI-format directory: smooth synthetic PNGs of the usual odd sizes, calib files
    with a P2 line, 15-field labels around the dataset's filter thresholds."""
    import io
    from PIL import Image
    files = {"ImageSets/train.txt": "\n".join("%06d" % i for i in ids).encode(), "ImageSets/val.txt": "\n".join("%06d" % i for i in ids).encode()}
    for k, idx in enumerate(ids):
        W, H = [(1242, 375), (1224, 370), (1238, 374), (1241, 376)][k % 4]
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx // 40 + yy // 40) % 2) * 200 + k * 10], -1).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="PNG")
        files["training/image_2/%06d.png" % idx] = buf.getvalue()
        fu = [721.5377, 707.0493, 718.856, 552.554][k % 4]
        P2 = [fu, 0.0, 609.5593 + k, 44.85728, 0.0, fu, 172.854 - k, 0.2163791, 0.0, 0.0, 1.0, 0.002745884]
        eye = [0.9999239, 0.00983776, -0.007445048, -0.009869795, 0.9999421, -0.004278459, 0.007402527, 0.004351614, 0.9999631]
        tr = [0.007533745, -0.9999714, -0.000616602, -0.004069766, 0.01480249, 0.0007280733, -0.9998902, -0.07631618, 0.9998621, 0.00752379, 0.01480755, -0.2717806]
        fmt = lambda name, v: name + ": " + " ".join("%.12e" % x for x in v)
        files["training/calib/%06d.txt" % idx] = "\n".join([fmt("P0", P2), fmt("P1", P2), fmt("P2", P2), fmt("P3", P2), fmt("R0_rect", eye),
                                                            fmt("Tr_velo_to_cam", tr), fmt("Tr_imu_to_velo", tr)]).encode() + b"\n"
        lines = []
        for _ in range(int(rng.integers(3, 9))):
            cls = ["Car", "Car", "Car", "Pedestrian", "Cyclist", "Van", "DontCare"][int(rng.integers(0, 7))]
            z = float(rng.uniform(1.5, 75))
            x, y = float(rng.uniform(-0.45, 0.45) * z), float(rng.uniform(1.4, 1.9))
            h, w, l = float(rng.normal(1.55, 0.1)), float(rng.normal(1.63, 0.1)), float(rng.normal(3.9, 0.4))
            ry = float(rng.uniform(-np.pi, np.pi))
            cu, cv = 609.5 + fu * x / z, 172.8 + fu * (y - h / 2) / z
            hw, hh = fu * l / z * float(rng.uniform(0.3, 0.55)), fu * h / z / 2
            off = float(rng.uniform(-0.6, 0.6)) * hw                       # some 3D centres fall outside their 2D box
            box = [max(cu - hw + off, 0.0), max(cv - hh, 0.0), min(cu + hw + off, W - 1.0), min(cv + hh, H - 1.0)]
            trunc = [0.0, 0.0, 0.2, 0.45, 0.7][int(rng.integers(0, 5))]
            occ = int(rng.integers(0, 4))
            if cls == "DontCare":
                lines.append("DontCare -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10" % tuple(box))
            else:
                lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" %
                             (cls, trunc, occ, ry - np.arctan2(x, z), box[0], box[1], box[2], box[3], h, w, l, x, y, z, ry))
        files["training/label_2/%06d.txt" % idx] = ("\n".join(lines) + "\n").encode()
    return files


def gen_kitti_dataset():
    """The reference KITTI_Dataset (kitti_dataset.py:27-489, with Object3d / Calibration / affine helpers of kitti_utils.py) on a
    small synthetic KITTI directory whose FILES are part of the fixture: val split (no augmentation) and train split with
    flip + crop augmentation under fixed numpy seeds, canonical object space on."""
    import importlib
    import tempfile
    _dataset_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    mod = importlib.import_module("lib.datasets.kitti.kitti_dataset")
    rng = np.random.default_rng(777)
    ids = [3, 7, 12, 25, 31, 40]
    files = _synthetic_kitti_files(rng, ids)
    out = {"file_names": np.array(sorted(files)), "ids": np.array(ids)}
    for i, name in enumerate(sorted(files)):
        out["file_%03d" % i] = np.frombuffer(files[name], dtype=np.uint8)
    cfg = {"root_dir": None, "writelist": ["Car", "Pedestrian"], "use_canonical_module": True, "canonical_focal_length": 500.0,
           "aug_crop": True, "random_flip": 0.5, "random_crop": 0.5, "scale": 0.05, "shift": 0.05, "depth_scale": "normal",
           "meanshape": False, "clip_2d": False}
    out["cfg_json"] = np.array(__import__("json").dumps({k: v for k, v in cfg.items() if k != "root_dir"}))
    with tempfile.TemporaryDirectory() as root:
        for name, blob in files.items():
            os.makedirs(os.path.dirname(os.path.join(root, name)), exist_ok=True)
            with open(os.path.join(root, name), "wb") as f:
                f.write(blob)
        cfg["root_dir"] = root
        for split, seeds in (("val", [0]), ("train", [11, 12, 13])):
            ds = mod.KITTI_Dataset(split, dict(cfg))
            for seed in seeds:
                for item in range(len(ds)):
                    np.random.seed(seed * 100 + item)
                    img, P2, targets, info = ds[item]
                    key = "%s_s%d_i%d__" % (split, seed, item)
                    out[key + "img_sub"] = np.ascontiguousarray(img[:, ::8, ::8])
                    out[key + "img_sum"] = np.float64(img.astype(np.float64).sum())
                    out[key + "P2"] = np.asarray(P2)
                    for k, v in targets.items():
                        out[key + "t_" + k] = np.asarray(v)
                    for k in ("img_id", "img_size", "bbox_downsample_ratio", "canonical_scale", "height_crop", "affine", "affine_inv",
                              "scale_depth", "flip"):
                        out[key + "info_" + k] = np.asarray(info[k])
    _save("kitti_dataset", **out)


def _cv2_color_shim():
    """cv2.cvtColor(float32 image, COLOR_BGR2HSV | COLOR_HSV2BGR) for pd.py:159-165.  OpenCV itself is absent from this image, so the
    shim restates OpenCV's float KERNEL as its source reads (imgproc color_hsv: RGB2HSV_f -- S = diff / (|V| + FLT_EPSILON), H scale
    (float)(60. / (diff + FLT_EPSILON)) -- and HSV2RGB_f: H * (6 / 360) wrapped into [0, 6), sector table), not the idealised formula
    of its documentation: the photometric fixture is pinned to OpenCV's published source, not to its binaries (written here
    independently of monosowa_amd/photometric.py, from the same source text)."""
    cv2 = sys.modules["cv2"]
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54
    eps = np.float32(np.finfo(np.float32).eps)

    def cvt(img, code):
        img = np.asarray(img, dtype=np.float32)
        f = np.float32
        if code == cv2.COLOR_BGR2HSV:
            out = np.empty_like(img)
            b, g, r = img[..., 0], img[..., 1], img[..., 2]
            v = np.maximum(r, np.maximum(g, b))
            vmin = np.minimum(r, np.minimum(g, b))
            diff = (v - vmin).astype(np.float32)
            out[..., 1] = diff / (np.abs(v) + eps)
            scale = (np.float64(60.0) / (diff + eps).astype(np.float64)).astype(np.float32)
            is_r, is_g = v == r, (v == g) & ~(v == r)
            h = np.where(is_r, (g - b) * scale, np.where(is_g, (b - r) * scale + f(120), (r - g) * scale + f(240))).astype(np.float32)
            out[..., 0] = np.where(h < 0, h + f(360), h)
            out[..., 2] = v
            return out
        assert code == cv2.COLOR_HSV2BGR
        h, s, v = img[..., 0], img[..., 1], img[..., 2]
        hh = (h * f(6.0 / 360.0)).astype(np.float32)
        hh = (hh - f(6) * np.floor(hh / f(6))).astype(np.float32)
        hh = np.where(hh >= 6, hh - f(6), hh).astype(np.float32)
        sec = np.floor(hh).astype(np.int64)
        fr = (hh - sec.astype(np.float32)).astype(np.float32)
        sec = np.where((sec < 0) | (sec > 5), 0, sec)
        tab = np.stack([v, v * (f(1) - s), v * (f(1) - s * fr), v * (f(1) - s * (f(1) - fr))], 0).astype(np.float32)
        sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # -> (b, g, r)
        idx = sector_data[sec]                                                                          # [..., 3]
        out = np.take_along_axis(tab, np.moveaxis(idx, -1, 0), 0)                                       # [3, ...]
        return np.moveaxis(out, 0, -1).astype(np.float32)
    cv2.cvtColor = cvt


def gen_kitti_dataset_pd():
    """The reference KITTI_Dataset with the dataset section of its shipped mixed-dataset config (checkpoints/
    best_kitti_k360_to_kitti/monodetr_kk360_05.yaml: aug_pd + aug_crop on, canonical focal length 1000) on the synthetic
    KITTI directory of gen_kitti_dataset (same files: they travel in kitti_dataset.npz), and the reference PhotometricDistort
    (pd.py:398-416) alone on a seeded float image."""
    import importlib
    import tempfile
    import yaml
    _dataset_shims()
    _cv2_color_shim()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    mod = importlib.import_module("lib.datasets.kitti.kitti_dataset")
    pdm = importlib.import_module("lib.datasets.kitti.pd")
    files = _synthetic_kitti_files(np.random.default_rng(777), [3, 7, 12, 25, 31, 40])
    shipped = yaml.safe_load(open(os.path.join(REF, "checkpoints", "best_kitti_k360_to_kitti", "monodetr_kk360_05.yaml")))["dataset"]
    cfg = {k: v for k, v in shipped.items() if k not in ("root_dir", "type", "batch_size", "train_split", "test_split")}
    assert cfg.get("aug_pd") is True and cfg.get("aug_crop") is True
    out = {"cfg_json": np.array(__import__("json").dumps(cfg))}
    # (a) the distortion alone: every branch combination shows up within 24 seeds
    rng = np.random.default_rng(5)
    image = (rng.uniform(0, 255, (24, 40, 3))).astype(np.float32)
    image[0, :4] = [[0, 0, 0], [255, 255, 255], [10, 10, 200], [200, 10, 10]]       # grey / saturated corner cases
    out["pd_image"] = image
    pd = pdm.PhotometricDistort()
    for seed in range(24):
        np.random.seed(1000 + seed)
        out["pd_out_%02d" % seed] = np.asarray(pd(image), dtype=np.float32)
    # (b) whole train samples
    with tempfile.TemporaryDirectory() as root:
        for name, blob in files.items():
            os.makedirs(os.path.dirname(os.path.join(root, name)), exist_ok=True)
            with open(os.path.join(root, name), "wb") as f:
                f.write(blob)
        ds = mod.KITTI_Dataset("train", dict(cfg, root_dir=root))
        for seed in (21, 22):
            for item in range(len(ds)):
                np.random.seed(seed * 100 + item)
                img, P2, targets, info = ds[item]
                key = "train_s%d_i%d__" % (seed, item)
                out[key + "img_sub"] = np.ascontiguousarray(img[:, ::8, ::8])
                out[key + "img_sum"] = np.float64(img.astype(np.float64).sum())
                out[key + "P2"] = np.asarray(P2)
                for k in ("boxes_3d", "depth", "mask_2d", "labels"):
                    out[key + "t_" + k] = np.asarray(targets[k])
                out[key + "info_flip"] = np.asarray(info["flip"])
    _save("kitti_dataset_pd", **out)


def importlib_misc():
    import importlib
    return importlib.import_module("utils.misc")


if __name__ == "__main__":
    which = sys.argv[1:] or ["msda"]
    for w in which:
        globals()["gen_" + w]()
