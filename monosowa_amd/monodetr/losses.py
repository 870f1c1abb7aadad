"""Loss building blocks of the criterion.

* ``sigmoid_focal_loss``  -- lib/losses/focal_loss.py:69-94 (RetinaNet focal loss on logits)
* ``softmax_focal_loss``  -- depth_predictor/ddn_loss/focalloss.py:12-136 (kornia-style, one-hot + eps)
* ``DDNLoss``             -- depth_predictor/ddn_loss/ddn_loss.py:12-127 + balancer.py:7-81:
  object-wise depth-map supervision: paint each 2D box with its centre depth (far boxes first),
  bin with LID to int64 targets, focal loss per pixel, foreground/background re-weighting.

Device handling follows the input tensors (the reference hard-codes 'cuda' at ddn_loss.py:32 and
monodetr.py:528, which is fatal on the CPU plumbing config).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def softmax_focal_loss(logits, target, alpha, gamma=2.0, eps=1e-6):
    """Per-pixel focal loss of a [B,C,H,W] logit map against int64 [B,H,W] targets ('none' reduction)."""
    soft = F.softmax(logits, dim=1)
    log_soft = F.log_softmax(logits, dim=1)
    one_hot = torch.zeros_like(logits).scatter_(1, target.unsqueeze(1), 1.0) + eps
    focal = -alpha * torch.pow(1.0 - soft, gamma) * log_soft
    return (one_hot * focal).sum(dim=1)


def lid_bin_indices(depth_map, depth_min=1e-3, depth_max=60, num_bins=80, target=True):
    """Linear-increasing-discretisation bin index (ddn_loss.py:85-100); out-of-range -> num_bins."""
    bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
    indices = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth_map - depth_min) / bin_size)
    if target:
        bad = (indices < 0) | (indices > num_bins) | (~torch.isfinite(indices))
        indices = indices.masked_fill(bad, num_bins).type(torch.int64)
    return indices


FUSED_DDN = True      # float32 CUDA logits: the depth-map loss as one kernel per direction


def _int_boxes(gt_boxes2d):
    """floor the top-left, ceil the bottom-right (ddn_loss.py:48-50, balancer.py:68-71)."""
    b = gt_boxes2d.clone()
    b[:, :2] = torch.floor(b[:, :2])
    b[:, 2:] = torch.ceil(b[:, 2:])
    return b.long()


class DDNLoss(nn.Module):
    def __init__(self, alpha=0.25, gamma=2.0, fg_weight=13, bg_weight=1, downsample_factor=1):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma
        self.fg_weight, self.bg_weight = fg_weight, bg_weight
        self.downsample_factor = downsample_factor

    @staticmethod
    def paint_boxes(shape, boxes_int, num_gt_per_img, fill_values=None, dtype=torch.bool, device=None):
        """Rasterise integer boxes image by image.  With ``fill_values`` (depths) boxes are painted far
        to near so that nearer objects overwrite farther ones (ddn_loss.py:56-62).  The box list is
        brought to the host once (it is a few dozen integers) instead of one sync per coordinate."""
        B, H, W = shape
        canvas = torch.zeros((B, H, W), dtype=dtype, device=device)
        boxes = boxes_int.tolist()
        depths = fill_values.tolist() if fill_values is not None else None
        start = 0
        for b, n in enumerate(num_gt_per_img):
            order = range(start, start + n)
            if depths is not None:   # descending depth, stable like torch.sort on the device
                order = sorted(order, key=lambda i: -depths[i])
            for i in order:
                u1, v1, u2, v2 = boxes[i]
                if depths is not None:
                    canvas[b, _sl(v1, v2, H), _sl(u1, u2, W)] = depths[i]
                else:
                    canvas[b, _sl(v1, v2, H), _sl(u1, u2, W)] = True
            start += n
        return canvas

    def forward_padded(self, depth_logits, boxes_padded, depth_padded, valid):
        """Same loss from padded per-image targets ([B,N,4] xyxy in depth-map pixels, [B,N], [B,N] bool) with no
        host round trip.  Equal depths aside (where the painted value is the same anyway) it reproduces
        ``forward`` exactly; downsample_factor must be 1 (the only value the reference uses)."""
        assert self.downsample_factor == 1
        B, _, H, W = depth_logits.shape
        if FUSED_DDN:
            from ..pointwise import ddn_loss, ddn_loss_supported
            if ddn_loss_supported(depth_logits, boxes_padded, depth_padded, valid):
                # rasterisation, LID binning, softmax focal loss and balancing: one HIP kernel each way (csrc/ddn_loss.hip)
                return ddn_loss(depth_logits, boxes_padded, depth_padded, valid, self.alpha, self.gamma, self.fg_weight, self.bg_weight)
        b = boxes_padded.clone()
        b[..., :2] = torch.floor(b[..., :2])
        b[..., 2:] = torch.ceil(b[..., 2:])
        depth_maps, fg = rasterize_boxes(b.long(), depth_padded, valid, H, W)
        target = lid_bin_indices(depth_maps, target=True)
        loss = softmax_focal_loss(depth_logits, target, self.alpha, self.gamma)
        loss = loss * (self.fg_weight * fg + self.bg_weight * (~fg))
        return (loss * fg).sum() / fg.numel() + (loss * (~fg)).sum() / fg.numel()

    def forward(self, depth_logits, gt_boxes2d, num_gt_per_img, gt_center_depth):
        B, _, H, W = depth_logits.shape
        boxes_int = _int_boxes(gt_boxes2d)
        depth_maps = self.paint_boxes((B, H, W), boxes_int, num_gt_per_img, gt_center_depth,
                                      dtype=depth_logits.dtype, device=depth_logits.device)
        target = lid_bin_indices(depth_maps, target=True)
        loss = softmax_focal_loss(depth_logits, target, self.alpha, self.gamma)
        # foreground / background balancing (balancer.py:24-52)
        # the reference divides the already integer-valued boxes (its first rasterisation edits them in place)
        boxes_fg = _int_boxes(boxes_int.to(gt_boxes2d.dtype) / self.downsample_factor)
        fg = self.paint_boxes((B, H, W), boxes_fg, num_gt_per_img, None, dtype=torch.bool, device=loss.device)
        weights = self.fg_weight * fg + self.bg_weight * (~fg)
        num_pixels = fg.numel()
        loss = loss * weights
        return (loss * fg).sum() / num_pixels + (loss * (~fg)).sum() / num_pixels


def _norm_slice(a, n):
    """Start/stop of ``[a:b]`` normalised the way Python/torch slicing does: negative values count from
    the end, everything is clamped into [0, n]."""
    a = torch.where(a < 0, a + n, a)
    return a.clamp(min=0, max=n)


def rasterize_boxes(boxes_int, depths, valid, H, W):
    """Loop-free equivalent of ``DDNLoss.paint_boxes`` for padded per-image boxes, fully on the device.
    boxes_int [B,N,4] int64 (u1,v1,u2,v2), depths [B,N], valid [B,N] bool ->
    depth_map [B,H,W] (painting far to near = the nearest covering box wins), fg_mask [B,H,W] bool."""
    u1, v1, u2, v2 = boxes_int.unbind(-1)
    ys = torch.arange(H, device=boxes_int.device).view(1, 1, H, 1)
    xs = torch.arange(W, device=boxes_int.device).view(1, 1, 1, W)
    y0, y1 = _norm_slice(v1, H)[..., None, None], _norm_slice(v2, H)[..., None, None]
    x0, x1 = _norm_slice(u1, W)[..., None, None], _norm_slice(u2, W)[..., None, None]
    cover = (ys >= y0) & (ys < y1) & (xs >= x0) & (xs < x1) & valid[..., None, None]          # [B,N,H,W]
    inf = torch.full((), float("inf"), dtype=depths.dtype, device=depths.device)
    nearest = torch.where(cover, depths[..., None, None], inf).amin(dim=1)
    fg = cover.any(dim=1)
    return torch.where(fg, nearest, torch.zeros((), dtype=depths.dtype, device=depths.device)), fg


def _sl(a, b, n):
    """``[a:b]`` exactly as tensor indexing treats it -- a negative start counts from the end, as it
    does in the reference when a box pokes out of the left/top border (ddn_loss.py:61)."""
    return slice(a, b)
