"""Detections from network outputs (reference: lib/helpers/decode_helper.py --
``extract_dets_from_outputs`` :58-111 on the device, ``decode_detections`` :8-55 in numpy).

The top-k / ``//`` / ``%`` / gather bookkeeping is integer work and is kept operation for operation
(north_star: bit-exact index/top-k bookkeeping).  Row layout of a detection (37 columns):
[cls, score, x2d, y2d, w2d, h2d, depth, heading(24), size3d(3), x3d, y3d, sigma].
"""
import numpy as np
import torch

from ..monodetr import box_ops

NUM_HEADING_BIN = 12


def class2angle(cls, residual, to_label_format=False):
    """Inverse of the 12-bin heading encoding (lib/datasets/utils.py:19-26)."""
    angle = cls * (2 * np.pi / float(NUM_HEADING_BIN)) + residual
    if to_label_format and angle > np.pi:
        angle = angle - 2 * np.pi
    return angle


def get_heading_angle(heading):
    cls = np.argmax(heading[0:12])
    return class2angle(cls, heading[12:24][cls], to_label_format=True)


class PinholeCalib:
    """The two camera helpers decoding needs (lib/datasets/kitti/kitti_utils.py:200-210,270-284),
    built from a 3x4 P2 matrix."""

    def __init__(self, P2):
        P2 = np.asarray(P2, dtype=np.float64)
        self.P2 = P2
        self.cu, self.cv = P2[0, 2], P2[1, 2]
        self.fu, self.fv = P2[0, 0], P2[1, 1]
        self.tx, self.ty = P2[0, 3] / (-self.fu), P2[1, 3] / (-self.fv)

    def img_to_rect(self, u, v, depth_rect):
        u, v, depth_rect = (np.asarray(a, dtype=np.float64) for a in (u, v, depth_rect))
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), depth_rect.reshape(-1, 1)), axis=1)

    def alpha2ry(self, alpha, u):
        ry = alpha + np.arctan2(u - self.cu, self.fu)
        if ry > np.pi:
            ry -= 2 * np.pi
        if ry < -np.pi:
            ry += 2 * np.pi
        return ry


DEVICE_KERNEL = True      # float32 CUDA outputs: one HIP launch (csrc/rotate_iou.hip, dets::extract_dets_kernel)


def extract_dets_from_outputs(outputs, K=50, topk=50):
    out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
    if DEVICE_KERNEL and out_logits.is_cuda and out_logits.dtype == torch.float32 and out_logits.shape[1] * out_logits.shape[2] <= 8192 \
            and all(outputs[k].dtype == torch.float32 for k in ("pred_boxes", "pred_angle", "pred_3d_dim", "pred_depth")):
        from ..kitti_eval import extract_dets_device
        return extract_dets_device(outputs, topk)
    batch, _, num_cls = out_logits.shape
    prob = out_logits.sigmoid()
    scores, topk_indexes = torch.topk(prob.view(batch, -1), topk, dim=1)
    topk_boxes = (topk_indexes // num_cls).unsqueeze(-1)      # query index
    labels = topk_indexes % num_cls                           # class index

    take = lambda t, width: torch.gather(t, 1, topk_boxes.repeat(1, 1, width))
    boxes = take(out_bbox, 6)
    heading = take(outputs["pred_angle"], 24)
    depth = take(outputs["pred_depth"][:, :, 0:1], 1)
    sigma = take(torch.exp(-outputs["pred_depth"][:, :, 1:2]), 1)
    size_3d = take(outputs["pred_3d_dim"], 3)

    xywh_2d = box_ops.box_xyxy_to_cxcywh(box_ops.box_cxcylrtb_to_xyxy(boxes))
    col = lambda t: t.reshape(batch, -1, 1)
    return torch.cat([col(labels), col(scores), col(xywh_2d[:, :, 0:1]), col(xywh_2d[:, :, 1:2]), xywh_2d[:, :, 2:4],
                      depth, heading, size_3d, col(boxes[:, :, 0:1]), col(boxes[:, :, 1:2]), sigma], dim=2)


def decode_detections(dets, info, calibs, cls_mean_size, threshold):
    """numpy: dets [B, K, 37] -> {img_id: [[cls, alpha, x1,y1,x2,y2, h,w,l, x,y,z, ry, score], ...]}.
    Array form of the reference's double loop (``decode_detections_loop`` below keeps the literal form; the two
    are compared in tests): one pass of vector arithmetic per batch instead of B x K Python iterations."""
    dets = np.asarray(dets)
    B, K, _ = dets.shape
    f64 = np.float64
    img_w = np.asarray(info["img_size"])[:, 0].astype(f64)[:, None]
    img_h = np.asarray(info["img_size"])[:, 1].astype(f64)[:, None]
    crop_h = img_h / np.asarray(info["height_crop"]).astype(f64).reshape(B, 1)
    padding = (img_h - crop_h) // 2
    cls_id = dets[:, :, 0].astype(np.int64)
    score = dets[:, :, 1]
    keep = ~(score < threshold)
    x = dets[:, :, 2] * img_w
    y = dets[:, :, 3] * crop_h + padding
    w = dets[:, :, 4] * img_w
    h = dets[:, :, 5] * crop_h
    depth = dets[:, :, 6] / np.asarray(info["canonical_scale"]).astype(f64).reshape(B, 1)
    dims = dets[:, :, 31:34] + np.asarray(cls_mean_size)[cls_id]
    x3d = dets[:, :, 34] * img_w
    y3d = dets[:, :, 35] * crop_h + padding
    cal = lambda name: np.array([getattr(c, name) for c in calibs], dtype=f64)[:, None]
    cu, cv, fu, fv, tx, ty = (cal(n) for n in ("cu", "cv", "fu", "fv", "tx", "ty"))
    loc_x = ((x3d - cu) * depth) / fu + tx
    loc_y = ((y3d - cv) * depth) / fv + ty + dims[:, :, 0] / 2
    heading = dets[:, :, 7:31]
    bin_id = np.argmax(heading[:, :, 0:12], axis=2)
    res = np.take_along_axis(heading[:, :, 12:24], bin_id[..., None], axis=2)[..., 0]
    alpha = bin_id * (2 * np.pi / float(NUM_HEADING_BIN)) + res
    alpha = np.where(alpha > np.pi, alpha - 2 * np.pi, alpha)
    ry = alpha + np.arctan2(x - cu, fu)
    ry = np.where(ry > np.pi, ry - 2 * np.pi, ry)
    ry = np.where(ry < -np.pi, ry + 2 * np.pi, ry)
    final = score * dets[:, :, -1]
    results = {}
    for i in range(B):
        rows = []
        for j in np.nonzero(keep[i])[0]:
            rows.append([int(cls_id[i, j]), alpha[i, j], x[i, j] - w[i, j] / 2, y[i, j] - h[i, j] / 2, x[i, j] + w[i, j] / 2,
                         y[i, j] + h[i, j] / 2] + dims[i, j].tolist() + [loc_x[i, j], loc_y[i, j], depth[i, j], ry[i, j], final[i, j]])
        results[info["img_id"][i]] = rows
    return results


def decode_detections_loop(dets, info, calibs, cls_mean_size, threshold):
    """The reference's formulation, detection by detection (decode_helper.py:8-55)."""
    results = {}
    for i in range(dets.shape[0]):
        preds = []
        img_w, img_h = info["img_size"][i][0], info["img_size"][i][1]
        crop_h = img_h / info["height_crop"][i]
        padding = (img_h - crop_h) // 2
        for j in range(dets.shape[1]):
            cls_id = int(dets[i, j, 0])
            score = dets[i, j, 1]
            if score < threshold:
                continue
            x = dets[i, j, 2] * img_w
            y = dets[i, j, 3] * crop_h + padding
            w = dets[i, j, 4] * img_w
            h = dets[i, j, 5] * crop_h
            bbox = [x - w / 2, y - h / 2, x + w / 2, y + h / 2]
            depth = dets[i, j, 6] / info["canonical_scale"][i]      # leave Canonical Object Space
            dimensions = dets[i, j, 31:34] + cls_mean_size[cls_id]
            x3d = dets[i, j, 34] * img_w
            y3d = dets[i, j, 35] * crop_h + padding
            locations = calibs[i].img_to_rect(x3d, y3d, depth).reshape(-1)
            locations[1] += dimensions[0] / 2
            alpha = get_heading_angle(dets[i, j, 7:31])
            ry = calibs[i].alpha2ry(alpha, x)
            score = score * dets[i, j, -1]
            preds.append([cls_id, alpha] + bbox + dimensions.tolist() + locations.tolist() + [ry, score])
        results[info["img_id"][i]] = preds
    return results
