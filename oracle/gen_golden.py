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


def importlib_misc():
    import importlib
    return importlib.import_module("utils.misc")


if __name__ == "__main__":
    which = sys.argv[1:] or ["msda"]
    for w in which:
        globals()["gen_" + w]()
