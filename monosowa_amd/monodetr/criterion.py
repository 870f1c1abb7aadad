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
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops
from .losses import DDNLoss, sigmoid_focal_loss
from .misc import accuracy, get_world_size, is_dist_avail_and_initialized


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, group_num=11, cfg=None,
                 depth_map_size=(80, 24)):
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
