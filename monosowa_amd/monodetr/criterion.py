"""Set-prediction criterion of MonoDETR (reference: lib/models/monodetr/monodetr.py --
``SetCriterion.__init__`` :308-340, losses :396-575, index helpers :1159-1169, ``get_loss``
:1171-1186, ``forward`` :1188-1230).

Loss dict keys and normalisation are the reference's: loss_ce, loss_bbox, loss_giou, loss_center,
loss_depth, loss_dim, loss_angle, loss_depth_map, loss_tfl, loss_mask (+ ``_i`` for the auxiliary
decoder layers), class_error / cardinality_error for logging.  The template-fitting / mask losses
(:342-394, :577-1157) are disabled in every shipped config (use_tfl / use_mask_loss False) and
depend on pytorch3d/open3d symbols the reference never imports; they return the same zero
placeholders here and raise if switched on.

``num_boxes`` is all-reduced over the data-parallel group (:1202-1206) so that N GPUs at
per-GPU batch b compute the same loss as one GPU at batch N*b.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from .losses import DDNLoss, sigmoid_focal_loss
from .misc import accuracy, get_world_size, is_dist_avail_and_initialized
from ..pointwise import focal_classification, focal_classification_supported, matched_losses, matched_losses_supported


def _paired_giou(a, b):
    """GIoU of matched xyxy boxes a[..., 4], b[..., 4] (the diagonal of box_ops.generalized_box_iou)."""
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    wh = (torch.min(a[..., 2:], b[..., 2:]) - torch.max(a[..., :2], b[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a + area_b - inter
    wh_c = (torch.max(a[..., 2:], b[..., 2:]) - torch.min(a[..., :2], b[..., :2])).clamp(min=0)
    area_c = wh_c[..., 0] * wh_c[..., 1]
    return inter / union - (area_c - union) / area_c


def _dev(values, dtype, device):
    """Small host values -> device WITHOUT blocking the host: on ROCm ``torch.tensor(list, device=cuda)`` and
    ``torch.as_tensor(list, device=cuda)`` wait for every queued kernel (214 ms behind a 214 ms queue,
    tools/h2d_probe.py), a staged ``.to(device, non_blocking=True)`` of a CPU tensor returns in 70 us."""
    return torch.as_tensor(np.asarray(values)).to(dtype).to(device, non_blocking=True)


FUSED_FOCAL = True       # classification side (focal sums, class / cardinality errors) as one HIP kernel each way
FUSED_MATCHED = True     # matched-pair losses through the HIP kernels on the GPU (False: the PyTorch formulation below)


class LossDict(dict):
    """The criterion's loss dictionary (same keys as the reference) that also remembers the [n_keys, n_layers] matrix its
    per-layer entries are views of, so that ``weighted_total`` is one multiply-and-sum instead of ~60 scalar kernels.
    With ``lazy`` set (name -> (row, layer)) the per-layer entries are created on first access: the ~30 select calls are
    host time directly behind the matcher's synchronisation, and the training loop reads them only when it logs."""
    mat = None          # [n_keys, NL] tensor; entry (i, l) is self[keys[i] + suffix(l)]
    keys_ = ()
    extras = ()         # names of the remaining differentiable scalars (loss_depth_map)
    lazy = None
    _zero = None

    def __missing__(self, key):
        if self.lazy is not None and key in self.lazy:
            i, l = self.lazy[key]
            if i >= 0:
                value = self.mat[i, l]
            else:                                    # loss_tfl / loss_mask: disabled losses, one shared zero
                if self._zero is None:
                    self._zero = torch.zeros((), device=self.mat.device, dtype=torch.float32, requires_grad=True)
                value = self._zero
            dict.__setitem__(self, key, value)
            return value
        raise KeyError(key)

    def _all_keys(self):
        seen = list(dict.keys(self))
        if self.lazy:
            have = set(seen)
            seen += [k for k in self.lazy if k not in have]
        return seen

    def __contains__(self, key):
        return dict.__contains__(self, key) or (self.lazy is not None and key in self.lazy)

    def __iter__(self):
        return iter(self._all_keys())

    def __len__(self):
        return len(self._all_keys())

    def keys(self):
        return self._all_keys()

    def items(self):
        return [(k, self[k]) for k in self._all_keys()]

    def values(self):
        return [self[k] for k in self._all_keys()]

    def get(self, key, default=None):
        return self[key] if key in self else default


_WEIGHT_CACHE = {}


def weighted_total(loss_dict, weight_dict):
    """``sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict)`` (trainer_helper.py:140-141).
    For the criterion's own LossDict: one [n_keys, n_layers] multiply-and-sum plus the depth-map term; the ~60 scalar
    multiplies and adds of the literal expression (and their ~60 backward kernels) sit right behind the matcher's host
    sync, where the GPU queue is empty and every launch is exposed."""
    mat = getattr(loss_dict, "mat", None)
    if mat is not None:
        NL = mat.shape[1]
        suffix = lambda l: "" if l == 0 else "_%d" % (l - 1)
        w_host = [[float(weight_dict.get(k + suffix(l), 0.0)) for l in range(NL)] for k in loss_dict.keys_]
        key = (loss_dict.keys_, NL, str(mat.device), tuple(map(tuple, w_host)))
        w = _WEIGHT_CACHE.get(key)
        if w is None:
            w = _WEIGHT_CACHE[key] = torch.tensor(w_host, dtype=mat.dtype).to(mat.device)
        total = (mat * w).sum()
        for k in loss_dict.extras:
            if k in weight_dict:
                total = total + loss_dict[k] * float(weight_dict[k])
        return total
    keys = [k for k in loss_dict if k in weight_dict]
    if not keys:
        return 0
    vals = torch.stack([loss_dict[k].reshape(()) for k in keys])
    w = _dev([float(weight_dict[k]) for k in keys], vals.dtype, vals.device) if vals.is_cuda else \
        torch.tensor([float(weight_dict[k]) for k in keys], dtype=vals.dtype)
    return torch.dot(vals, w)


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, group_num=11, cfg=None,
                 depth_map_size=(80, 24), fast=True):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        self.ddn_loss = DDNLoss()
        self.group_num = group_num
        # width, height of the stride-16 depth map; the reference hard-codes [80, 24, 80, 24] (:528)
        self.depth_map_size = tuple(depth_map_size)
        # fast=True: all decoder layers matched and scored together with flat indices (one device->host
        # copy per step); fast=False: the layer-by-layer, image-by-image formulation of the reference.
        self.fast = fast
        self.use_tfl = bool(cfg["use_tfl"]) if cfg is not None else False
        self.use_mask_loss = bool(cfg["use_mask_loss"]) if cfg is not None else False
        self.mask_loss = cfg["mask_loss"] if cfg is not None else "DICE"
        if self.use_tfl or self.use_mask_loss:
            raise NotImplementedError("use_tfl / use_mask_loss need pytorch3d + open3d rendering that the reference "
                                      "itself does not import; they are off in every shipped config")

    # ------------------------------------------------------------------ index helpers
    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    @staticmethod
    def _get_tgt_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        tgt_idx = torch.cat([tgt for (_, tgt) in indices])
        return batch_idx, tgt_idx

    @staticmethod
    def _matched(targets, indices, key, cols=None):
        parts = [t[key][i] if cols is None else t[key][:, cols][i] for t, (_, i) in zip(targets, indices)]
        return torch.cat(parts, dim=0)

    # ------------------------------------------------------------------ individual losses
    def loss_labels(self, outputs, targets, indices, num_boxes, log=True, info=None):
        src_logits = outputs["pred_logits"]
        idx = self._get_src_permutation_idx(indices)
        target_classes_o = self._matched(targets, indices, "labels")
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        target_classes[idx] = target_classes_o.squeeze().long()
        onehot = torch.zeros([src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1],
                             dtype=src_logits.dtype, device=src_logits.device)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        loss_ce = sigmoid_focal_loss(src_logits, onehot[:, :, :-1], num_boxes, alpha=self.focal_alpha, gamma=2) \
            * src_logits.shape[1]
        losses = {"loss_ce": loss_ce}
        if log:
            losses["class_error"] = 100 - accuracy(src_logits[idx], target_classes_o)[0]
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes, info=None):
        pred_logits = outputs["pred_logits"]
        tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets], device=pred_logits.device)
        card_pred = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1)
        return {"cardinality_error": F.l1_loss(card_pred.float(), tgt_lengths.float())}

    def loss_3dcenter(self, outputs, targets, indices, num_boxes, info=None):
        idx = self._get_src_permutation_idx(indices)
        src = outputs["pred_boxes"][:, :, 0:2][idx]
        tgt = self._matched(targets, indices, "boxes_3d", slice(0, 2))
        return {"loss_center": F.l1_loss(src, tgt, reduction="none").sum() / num_boxes}

    def loss_boxes(self, outputs, targets, indices, num_boxes, info=None):
        idx = self._get_src_permutation_idx(indices)
        src_lrtb = outputs["pred_boxes"][:, :, 2:6][idx]
        tgt_lrtb = self._matched(targets, indices, "boxes_3d", slice(2, 6))
        losses = {"loss_bbox": F.l1_loss(src_lrtb, tgt_lrtb, reduction="none").sum() / num_boxes}
        src_boxes = outputs["pred_boxes"][idx]
        tgt_boxes = self._matched(targets, indices, "boxes_3d")
        giou = torch.diag(box_ops.generalized_box_iou(box_ops.box_cxcylrtb_to_xyxy(src_boxes),
                                                      box_ops.box_cxcylrtb_to_xyxy(tgt_boxes), check=False))
        losses["loss_giou"] = (1 - giou).sum() / num_boxes
        return losses

    def loss_depths(self, outputs, targets, indices, num_boxes, info=None):
        idx = self._get_src_permutation_idx(indices)
        src = outputs["pred_depth"][idx]
        tgt = self._matched(targets, indices, "depth").squeeze()
        depth, log_var = src[:, 0], src[:, 1]
        loss = 1.4142 * torch.exp(-log_var) * torch.abs(depth - tgt) + log_var     # Laplacian aleatoric
        return {"loss_depth": loss.sum() / num_boxes}

    def loss_dims(self, outputs, targets, indices, num_boxes, info=None):
        idx = self._get_src_permutation_idx(indices)
        src = outputs["pred_3d_dim"][idx]
        tgt = self._matched(targets, indices, "size_3d")
        dim_loss = torch.abs(src - tgt) / tgt.clone().detach()
        with torch.no_grad():
            compensation = F.l1_loss(src, tgt) / dim_loss.mean()
        return {"loss_dim": (dim_loss * compensation).sum() / num_boxes}

    def loss_angles(self, outputs, targets, indices, num_boxes, info=None):
        idx = self._get_src_permutation_idx(indices)
        heading = outputs["pred_angle"][idx].view(-1, 24)
        cls_t = self._matched(targets, indices, "heading_bin").view(-1).long()
        res_t = self._matched(targets, indices, "heading_res").view(-1)
        cls_loss = F.cross_entropy(heading[:, 0:12], cls_t, reduction="none")
        onehot = torch.zeros(cls_t.shape[0], 12, device=heading.device, dtype=heading.dtype).scatter_(
            dim=1, index=cls_t.view(-1, 1), value=1)
        res_pred = torch.sum(heading[:, 12:24] * onehot, 1)
        reg_loss = F.l1_loss(res_pred, res_t, reduction="none")
        return {"loss_angle": (cls_loss + reg_loss).sum() / num_boxes}

    def loss_depth_map(self, outputs, targets, indices, num_boxes, info=None):
        logits = outputs["pred_depth_map_logits"]
        num_gt_per_img = [len(t["boxes"]) for t in targets]
        w, h = self.depth_map_size
        scale = torch.tensor([w, h, w, h], device=logits.device, dtype=logits.dtype)
        boxes = box_ops.box_cxcywh_to_xyxy(torch.cat([t["boxes"] for t in targets], dim=0) * scale)
        centre_depth = torch.cat([t["depth"] for t in targets], dim=0).squeeze(dim=1)
        return {"loss_depth_map": self.ddn_loss(logits, boxes, num_gt_per_img, centre_depth)}

    def loss_tfl(self, outputs, targets, indices, num_boxes, info=None):
        dev = outputs["pred_logits"].device
        return {"loss_tfl": torch.tensor(0., device=dev, dtype=torch.float32, requires_grad=True),
                "loss_mask": torch.tensor(0., device=dev, dtype=torch.float32, requires_grad=True)}

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        loss_map = {"labels": self.loss_labels, "cardinality": self.loss_cardinality, "boxes": self.loss_boxes,
                    "depths": self.loss_depths, "dims": self.loss_dims, "angles": self.loss_angles,
                    "center": self.loss_3dcenter, "depth_map": self.loss_depth_map, "tfl": self.loss_tfl}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_boxes, **kwargs)

    def forward(self, outputs, targets, mask_dict=None, info=None):
        if self.fast:
            return self.forward_fast(outputs, targets)
        return self.forward_layerwise(outputs, targets, mask_dict, info)

    # ------------------------------------------------------------------ batched formulation
    def _num_boxes(self, targets, group_num, device):
        n = float(sum(len(t["labels"]) for t in targets) * group_num)
        if is_dist_avail_and_initialized():       # stays on the device: dividing by a tensor needs no sync
            t = _dev([n], torch.float, device)
            torch.distributed.all_reduce(t)
            return torch.clamp(t / get_world_size(), min=1)[0]
        return max(n / get_world_size(), 1.0)

    def forward_fast(self, outputs, targets):
        """Same losses as ``forward_layerwise`` (same keys, same normalisation); the matching of all decoder
        layers is one cost pass + one host copy + one native call, and every loss is evaluated for all layers
        at once through flat (layer, batch, query) / target index tensors."""
        layers = [{k: v for k, v in outputs.items() if k != "aux_outputs"}] + list(outputs.get("aux_outputs", []))
        NL = len(layers)
        group_num = self.group_num if self.training else 1
        logits = torch.stack([o["pred_logits"] for o in layers])          # [NL,B,Q,C]
        boxes = torch.stack([o["pred_boxes"] for o in layers])            # [NL,B,Q,6]
        dims = torch.stack([o["pred_3d_dim"] for o in layers])
        depth = torch.stack([o["pred_depth"] for o in layers])
        angle = torch.stack([o["pred_angle"] for o in layers])
        _, B, Q, C = logits.shape
        dev = logits.device
        sizes = [len(t["labels"]) for t in targets]
        T = sum(sizes)
        flat_keys = ("labels", "boxes_3d", "boxes", "depth", "size_3d", "heading_bin", "heading_res")
        flat = getattr(targets, "flat", None)              # prepare_targets hands the batch-flat tensors along (7 launches fewer)
        if flat is None or any(k not in flat for k in flat_keys) or flat["labels"].shape[0] != T:
            flat = {k: torch.cat([t[k] for t in targets], dim=0) for k in flat_keys}
        num_boxes = self._num_boxes(targets, group_num, dev)

        pending = self.matcher.match_layers_begin(logits, boxes, flat, sizes, group_num)      # cost pass + async D2H

        # depth map (final layer only): padded per-image boxes, rasterised on the device.  Independent of the matching,
        # so it is enqueued between the two halves of the matcher: the GPU works on it while the host waits for the
        # cost blocks and solves the assignments.
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        maxn = max(sizes) if sizes else 0
        w, h = self.depth_map_size
        if maxn:
            slot = np.minimum(offs[:, None] + np.arange(maxn)[None, :], max(T - 1, 0))
            valid = np.arange(maxn)[None, :] < np.asarray(sizes)[:, None]
            slot_t = torch.as_tensor(slot, dtype=torch.int64).to(dev, non_blocking=True)
            valid_t = torch.as_tensor(valid).to(dev, non_blocking=True)
            scale = _dev([w, h, w, h], logits.dtype, dev)
            boxes2d = box_ops.box_cxcywh_to_xyxy(flat["boxes"] * scale)[slot_t]
            depth2d = flat["depth"].squeeze(1)[slot_t]
        else:
            valid_t = torch.zeros((B, 1), dtype=torch.bool, device=dev)
            boxes2d = torch.zeros((B, 1, 4), device=dev, dtype=logits.dtype)
            depth2d = torch.zeros((B, 1), device=dev, dtype=logits.dtype)
        loss_depth_map = self.ddn_loss.forward_padded(outputs["pred_depth_map_logits"], boxes2d, depth2d, valid_t)

        # everything the fused tail needs that does not depend on the assignment is made ready BEFORE the wait: behind it the
        # GPU queue is empty and every microsecond of host time is step time (measured: 1.05 ms from indices to total)
        fused_tail = FUSED_FOCAL and FUSED_MATCHED and T > 0 and logits.is_cuda and logits.dtype == torch.float32
        if fused_tail:
            f32, i64 = torch.float32, torch.int64
            prep = [t.contiguous() for t in (logits, boxes, depth, dims, angle)]
            prep_t = [flat["boxes_3d"].to(f32).contiguous(), flat["depth"].reshape(-1).to(f32).contiguous(), flat["size_3d"].to(f32).contiguous(),
                      flat["heading_bin"].reshape(-1).to(i64).contiguous(), flat["heading_res"].reshape(-1).to(f32).contiguous()]
            labels64 = flat["labels"].to(i64).contiguous()
            sizes_dev = _dev(sizes, f32, dev)
            nb = num_boxes if torch.is_tensor(num_boxes) else float(num_boxes)
            # rows of the loss matrix: focal sum, class error, cardinality error, six matched-pair sums; divided by num_boxes
            # where the reference does (a [9, 1] column, 1 for the two logging entries)
            inv = (1.0 / nb) if not torch.is_tensor(nb) else None
            scale_col = _dev([inv, 1.0, 1.0] + [inv] * 6, f32, dev).view(9, 1) if inv is not None else \
                torch.cat([1.0 / nb.reshape(1), nb.new_ones(2), (1.0 / nb.reshape(1)).expand(6)]).view(9, 1)

        idx_host = self.matcher.match_layers_end_flat(pending)                 # [3, NL, K] int64 (host: numpy, or a pinned tensor)
        K = idx_host.shape[2]
        # from the matcher's pinned buffer the copy is queued behind the depth-map kernels still running; from pageable memory
        # the host would wait for them here, with the whole backward still to enqueue
        idx = (idx_host if torch.is_tensor(idx_host) else torch.from_numpy(idx_host)).to(dev, non_blocking=True)
        if fused_tail and K > 0 and focal_classification_supported(logits, idx):
            # the whole criterion behind the matching in four launches: classification side + matched-pair losses, forward
            # and backward (csrc/matched_losses.hip), the loss matrix in two more
            from ..pointwise import _FocalClassification, _MatchedLosses
            cls3 = _FocalClassification.apply(prep[0], idx, labels64, sizes_dev, float(self.focal_alpha), 2.0)
            sums = _MatchedLosses.apply(prep[1], prep[2], prep[3], prep[4], idx, *prep_t)
            mat = torch.cat([cls3, sums], 1).t() * scale_col                     # [9, NL]
            keys = ("loss_ce", "class_error", "cardinality_error", "loss_center", "loss_bbox", "loss_giou", "loss_depth", "loss_dim",
                    "loss_angle")
            return self._finish(keys, NL, dev, loss_depth_map, mat=mat)
        b_idx, q_idx, t_idx = idx[0], idx[1], idx[2]
        l_idx = torch.arange(NL, device=dev).view(NL, 1).expand(NL, K)
        take = lambda t: t[l_idx, b_idx, q_idx]                             # [NL,K,...] matched predictions
        per_layer = {}

        # labels (focal), class error, cardinality
        tgt_cls = flat["labels"].long()[t_idx] if T else torch.zeros((NL, 0), dtype=torch.int64, device=dev)
        target_classes = torch.full((NL, B, Q), self.num_classes, dtype=torch.int64, device=dev)
        target_classes[l_idx, b_idx, q_idx] = tgt_cls
        onehot = torch.zeros((NL, B, Q, C + 1), dtype=logits.dtype, device=dev).scatter_(3, target_classes.unsqueeze(-1), 1)[..., :-1]
        prob = logits.sigmoid()
        ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
        p_t = prob * onehot + (1 - prob) * (1 - onehot)
        focal = ce * ((1 - p_t) ** 2)
        if self.focal_alpha >= 0:
            focal = (self.focal_alpha * onehot + (1 - self.focal_alpha) * (1 - onehot)) * focal
        per_layer["loss_ce"] = focal.mean(2).sum((1, 2)) / num_boxes * Q
        matched_logits = take(logits)
        if K:
            correct = matched_logits.argmax(-1).eq(tgt_cls).float().sum(1) * (100.0 / K)
        else:
            correct = torch.zeros(NL, device=dev)
        per_layer["class_error"] = 100 - correct
        tgt_lengths = _dev(sizes, torch.float, dev)
        card_pred = (logits.argmax(-1) != C - 1).sum(2).float()
        per_layer["cardinality_error"] = (card_pred - tgt_lengths).abs().mean(1)

        if FUSED_MATCHED and matched_losses_supported(boxes, idx):
            # all six matched-pair losses of all layers: one HIP launch forward, one backward (csrc/matched_losses.hip)
            sums = matched_losses(boxes, depth, dims, angle, idx, flat["boxes_3d"], flat["depth"], flat["size_3d"],
                                  flat["heading_bin"], flat["heading_res"]) / num_boxes
            for j, k in enumerate(("loss_center", "loss_bbox", "loss_giou", "loss_depth", "loss_dim", "loss_angle")):
                per_layer[k] = sums[:, j]
            return self._finish(per_layer, NL, dev, loss_depth_map)
        # boxes: 3D-centre L1, l/r/t/b L1, GIoU of matched pairs
        src_box, tgt_box = take(boxes), flat["boxes_3d"][t_idx]
        per_layer["loss_center"] = (src_box[..., 0:2] - tgt_box[..., 0:2]).abs().sum((1, 2)) / num_boxes
        per_layer["loss_bbox"] = (src_box[..., 2:6] - tgt_box[..., 2:6]).abs().sum((1, 2)) / num_boxes
        giou = _paired_giou(box_ops.box_cxcylrtb_to_xyxy(src_box), box_ops.box_cxcylrtb_to_xyxy(tgt_box))
        per_layer["loss_giou"] = (1 - giou).sum(1) / num_boxes
        # depth (Laplacian aleatoric uncertainty)
        src_d, tgt_d = take(depth), flat["depth"][t_idx].squeeze(-1)
        per_layer["loss_depth"] = (1.4142 * torch.exp(-src_d[..., 1]) * (src_d[..., 0] - tgt_d).abs() + src_d[..., 1]).sum(1) / num_boxes
        # 3D size (dimension-aware L1 with the per-layer compensation weight)
        src_s, tgt_s = take(dims), flat["size_3d"][t_idx]
        l1 = (src_s - tgt_s).abs()
        dim_loss = l1 / tgt_s.detach()
        with torch.no_grad():
            comp = l1.mean((1, 2)) / dim_loss.mean((1, 2))
        per_layer["loss_dim"] = (dim_loss * comp.view(NL, 1, 1)).sum((1, 2)) / num_boxes
        # heading: 12-bin classification + residual of the target bin
        heading = take(angle)
        cls_t = flat["heading_bin"][t_idx].view(NL, K).long()
        res_t = flat["heading_res"][t_idx].view(NL, K)
        cls_loss = F.cross_entropy(heading[..., 0:12].reshape(NL * K, 12), cls_t.reshape(-1), reduction="none").view(NL, K)
        res_pred = torch.gather(heading[..., 12:24], 2, cls_t.unsqueeze(-1)).squeeze(-1)
        per_layer["loss_angle"] = (cls_loss + (res_pred - res_t).abs()).sum(1) / num_boxes

        return self._finish(per_layer, NL, dev, loss_depth_map)

    @staticmethod
    def _finish(per_layer, NL, dev, loss_depth_map, mat=None):
        losses = LossDict()
        keys = tuple(per_layer) if mat is None else per_layer
        if mat is None:
            mat = torch.stack([per_layer[k] for k in keys])                  # [n_keys, NL]
        lazy = {}
        for l in range(NL):
            suffix = "" if l == 0 else "_%d" % (l - 1)
            for i, k in enumerate(keys):
                lazy[k + suffix] = (i, l)
            lazy["loss_tfl" + suffix] = (-1, l)
            lazy["loss_mask" + suffix] = (-1, l)
        losses.lazy = lazy
        dict.__setitem__(losses, "loss_depth_map", loss_depth_map)
        losses.mat, losses.keys_, losses.extras = mat, keys, ("loss_depth_map",)
        return losses

    # ------------------------------------------------------------------ reference formulation
    def forward_layerwise(self, outputs, targets, mask_dict=None, info=None):
        outputs_without_aux = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        group_num = self.group_num if self.training else 1
        indices = self.matcher(outputs_without_aux, targets, group_num=group_num)

        num_boxes = sum(len(t["labels"]) for t in targets) * group_num
        num_boxes = torch.as_tensor([num_boxes], dtype=torch.float, device=next(iter(outputs.values())).device)
        if is_dist_avail_and_initialized():
            torch.distributed.all_reduce(num_boxes)
        num_boxes = torch.clamp(num_boxes / get_world_size(), min=1).item()

        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, indices, num_boxes, info=info))
        if "aux_outputs" in outputs:
            for i, aux in enumerate(outputs["aux_outputs"]):
                indices = self.matcher(aux, targets, group_num=group_num)
                for loss in self.losses:
                    if loss == "depth_map":     # the depth map has no per-layer prediction
                        continue
                    l_dict = self.get_loss(loss, aux, targets, indices, num_boxes, info=info)
                    losses.update({k + f"_{i}": v for k, v in l_dict.items()})
        return losses
