"""Hungarian matcher with per-group assignment (reference: lib/models/monodetr/matcher.py:14-112).

Cost = cost_class * focal-style class cost + cost_3dcenter * L1(projected 3D centre)
     + cost_bbox * L1(l,r,t,b) + cost_giou * (-GIoU); queries are split into ``group_num`` groups
and each group is matched to the targets independently (one-to-many supervision in training).
The assignment itself (scipy ``linear_sum_assignment`` on the host, as in the reference :87-103)
is index bookkeeping and must be bit-exact; one device->host copy of the cost matrix per call.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from .box_ops import box_cxcylrtb_to_xyxy, generalized_box_iou
from ..pointwise import match_cost_blocks, match_cost_supported


def _pairwise_l1(a, b):
    d = (a[:, None, :] - b[None, :, :]).abs()
    out = d[..., 0]
    for i in range(1, d.shape[-1]):
        out = out + d[..., i]
    return out


DEVICE_LSAP = True     # match_layers: the assignments on the device (csrc/lsap_device.hip: scipy's algorithm, arithmetic and tie-breaks,
                       # one wavefront per problem) -- no device -> host copy of the cost blocks, no host wait: the train step runs
                       # without a single host synchronisation.  False: the host solver behind one copy + one wait (csrc/lsap.cpp)
FUSED_COST = True      # ... from one HIP launch (pointwise.match_cost_blocks) instead of ~70 elementwise ones; same floats
BLOCK_COST = True      # match_layers: per-image cost blocks only (False: full cross matrix + gather, as the reference)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_3dcenter: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        self.cost_class = cost_class
        self.cost_3dcenter = cost_3dcenter
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost_matrix(self, outputs, targets):
        bs, num_queries = outputs["pred_boxes"].shape[:2]
        out_prob = outputs["pred_logits"].flatten(0, 1).sigmoid()
        tgt_ids = torch.cat([v["labels"] for v in targets]).long()
        alpha, gamma = 0.25, 2.0
        neg_cost = (1 - alpha) * (out_prob ** gamma) * (-(1 - out_prob + 1e-8).log())
        pos_cost = alpha * ((1 - out_prob) ** gamma) * (-(out_prob + 1e-8).log())
        cost_class = pos_cost[:, tgt_ids] - neg_cost[:, tgt_ids]

        out_box = outputs["pred_boxes"].flatten(0, 1)
        tgt_box = torch.cat([v["boxes_3d"] for v in targets])
        # pairwise L1 distances; the same left-to-right sum as the reference's torch.cdist(p=1) evaluates for 2 and
        # 4 components, without cdist's ~1.5 ms kernel on a [B*Q, 2] x [T, 2] problem
        cost_3dcenter = _pairwise_l1(out_box[:, 0:2], tgt_box[:, 0:2])
        cost_bbox = _pairwise_l1(out_box[:, 2:6], tgt_box[:, 2:6])
        cost_giou = -generalized_box_iou(box_cxcylrtb_to_xyxy(out_box), box_cxcylrtb_to_xyxy(tgt_box), check=False)
        C = self.cost_bbox * cost_bbox + self.cost_3dcenter * cost_3dcenter + self.cost_class * cost_class \
            + self.cost_giou * cost_giou
        return C.view(bs, num_queries, -1)

    @torch.no_grad()
    def cost_blocks(self, pred_logits, pred_boxes, tgt_ids, tgt_box):
        """The per-image diagonal blocks of ``cost_matrix`` only: pred_logits [NL,B,Q,C], pred_boxes [NL,B,Q,6] against
        each image's own (padded) targets tgt_ids [B,N], tgt_box [B,N,6] -> [NL,B,Q,N].  Element for element the same
        operations in the same order as ``cost_matrix`` (so the same floats and the same assignments), on 1/B of the
        elements: the full [B*Q, T] cross matrix costs 0.5 ms of GPU time on the critical path before the step's sync."""
        out_prob = pred_logits.sigmoid()
        alpha, gamma = 0.25, 2.0
        neg_cost = (1 - alpha) * (out_prob ** gamma) * (-(1 - out_prob + 1e-8).log())
        pos_cost = alpha * ((1 - out_prob) ** gamma) * (-(out_prob + 1e-8).log())
        NL, B, Q, _ = pred_logits.shape
        ids = tgt_ids.long().view(1, B, 1, -1).expand(NL, B, Q, -1)
        cost_class = pos_cost.gather(3, ids) - neg_cost.gather(3, ids)
        a, b = pred_boxes.unsqueeze(3), tgt_box.view(1, B, 1, -1, 6)                  # [NL,B,Q,1,6], [1,B,1,N,6]

        def l1(lo, hi):
            d = (a[..., lo:hi] - b[..., lo:hi]).abs()
            out = d[..., 0]
            for i in range(1, hi - lo):
                out = out + d[..., i]
            return out
        cost_3dcenter, cost_bbox = l1(0, 2), l1(2, 6)
        xa, xb = box_cxcylrtb_to_xyxy(a), box_cxcylrtb_to_xyxy(b)
        area1 = (xa[..., 2] - xa[..., 0]) * (xa[..., 3] - xa[..., 1])
        area2 = (xb[..., 2] - xb[..., 0]) * (xb[..., 3] - xb[..., 1])
        wh = (torch.min(xa[..., 2:], xb[..., 2:]) - torch.max(xa[..., :2], xb[..., :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - inter
        iou = inter / union
        whc = (torch.max(xa[..., 2:], xb[..., 2:]) - torch.min(xa[..., :2], xb[..., :2])).clamp(min=0)
        area = whc[..., 0] * whc[..., 1]
        cost_giou = -(iou - (area - union) / area)
        return self.cost_bbox * cost_bbox + self.cost_3dcenter * cost_3dcenter + self.cost_class * cost_class \
            + self.cost_giou * cost_giou

    @torch.no_grad()
    def forward(self, outputs, targets, group_num=11):
        """-> list (per image) of (query_idx int64, target_idx int64), concatenated over groups."""
        bs, num_queries = outputs["pred_boxes"].shape[:2]
        C = self.cost_matrix(outputs, targets).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        g_q = num_queries // group_num
        indices = None
        for g, C_g in enumerate(C.split(g_q, dim=1)[:group_num]):
            ind_g = [linear_sum_assignment(c[i]) for i, c in enumerate(C_g.split(sizes, -1))]
            if indices is None:
                indices = ind_g
            else:
                indices = [(np.concatenate([a[0], b[0] + g_q * g]), np.concatenate([a[1], b[1]]))
                           for a, b in zip(indices, ind_g)]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in indices]


    @torch.no_grad()
    def match_layers(self, pred_logits, pred_boxes, flat_targets, sizes, group_num):
        """All decoder layers at once: pred_logits [NL,B,Q,C], pred_boxes [NL,B,Q,6]; flat_targets holds the
        concatenated 'labels' [T] and 'boxes_3d' [T,6]; sizes = targets per image (host ints).
        One cost pass (same ops as ``cost_matrix``, so the same floats), one device->host copy of the
        per-image diagonal blocks, one native call for every (layer, image, group) assignment.
        -> list over layers of list over images of (query_idx, target_idx) int64 numpy arrays."""
        handle = self.match_layers_begin(pred_logits, pred_boxes, flat_targets, sizes, group_num)
        return self.match_layers_end(handle)

    @torch.no_grad()
    def match_layers_begin(self, pred_logits, pred_boxes, flat_targets, sizes, group_num):
        """First half of ``match_layers``: enqueues the cost pass and the device->host copy of the per-image blocks
        (pinned buffer, asynchronous) and returns a handle.  GPU work enqueued between this call and
        ``match_layers_end`` overlaps the host's wait and the assignment solve."""
        NL, B, Q, _ = pred_logits.shape
        T = int(sum(sizes))
        if T == 0:
            return None, NL, B, Q, sizes, group_num
        maxn = max(sizes)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        cols = np.minimum(offs[:, None] + np.arange(maxn)[None, :], T - 1)            # [B, maxn], clamped padding
        cols = torch.as_tensor(cols, dtype=torch.int64).to(pred_logits.device, non_blocking=True)
        if BLOCK_COST and FUSED_COST and match_cost_supported(pred_logits, pred_boxes):
            # the same floats from one HIP launch (csrc/matched_losses.hip: match_cost_kernel)
            blocks = match_cost_blocks(pred_logits, pred_boxes, flat_targets["labels"], flat_targets["boxes_3d"], cols, self.cost_class,
                                       self.cost_3dcenter, self.cost_bbox, self.cost_giou)
        elif BLOCK_COST:
            blocks = self.cost_blocks(pred_logits, pred_boxes, flat_targets["labels"][cols], flat_targets["boxes_3d"][cols])
        else:
            C = self.cost_matrix({"pred_logits": pred_logits.flatten(0, 1), "pred_boxes": pred_boxes.flatten(0, 1)},
                                 [flat_targets]).view(NL, B, Q, T)
            blocks = torch.gather(C, 3, cols.view(1, B, 1, maxn).expand(NL, B, Q, maxn))
        if blocks.is_cuda and DEVICE_LSAP:
            from ..pointwise import device_lsap_match_flat, device_lsap_supported
            blocks = blocks.contiguous()
            if device_lsap_supported(blocks, sizes, group_num):
                self.check_device_status()                      # an EARLIER call's flag, if its copy has landed: never a wait
                if getattr(self, "_status", None) is None or self._status.device != blocks.device:
                    self._status = torch.zeros((), dtype=torch.int32, device=blocks.device)
                    self._status_host = torch.zeros((), dtype=torch.int32).pin_memory()
                idx = device_lsap_match_flat(blocks, sizes, group_num, self._status)
                self._status_host.copy_(self._status, non_blocking=True)
                self._status_event = torch.cuda.Event()
                self._status_event.record()
                return ("device", idx), NL, B, Q, sizes, group_num
        if blocks.is_cuda:
            key = (tuple(blocks.shape), blocks.dtype)
            if getattr(self, "_pinned_key", None) != key:
                self._pinned, self._pinned_key = torch.empty(blocks.shape, dtype=blocks.dtype).pin_memory(), key
            self._pinned.copy_(blocks, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            return (self._pinned, done), NL, B, Q, sizes, group_num
        return (blocks, None), NL, B, Q, sizes, group_num

    def check_device_status(self, block=False):
        """The device solver cannot raise from a kernel: it ORs a flag into a status word whose copy to pinned memory is queued
        behind it.  Called at the start of the next matching (and by whoever wants certainty, with ``block=True``): raises what
        scipy raises for a cost matrix with NaN / -inf entries or without a feasible assignment."""
        ev = getattr(self, "_status_event", None)
        if ev is None:
            return
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        self._status_event = None
        word = int(self._status_host)
        if word & 3:
            self._status.zero_()
            if word & 1:
                raise ValueError("cost matrix is infeasible")
            # bit 1 (value 2): a problem exceeded the device solver's tables and got placeholder identity pairs instead of an assignment
            raise ValueError("assignment problem exceeds the device solver's capacity (identity pairs were emitted)")

    @torch.no_grad()
    def match_layers_end_flat(self, handle):
        """Second half of ``match_layers`` returning the flat int64 index array [3, NL, K] (image, query, flat target)."""
        from .. import lsap
        payload, NL, B, Q, sizes, group_num = handle
        if payload is not None and isinstance(payload[0], str):
            return payload[1]                                                         # already on the device: nothing to wait for
        if payload is not None and lsap.available():
            blocks, done = payload
            if done is not None:
                done.synchronize()                                                    # the step's one host sync
            return lsap.match_flat(blocks.numpy(), np.asarray(sizes, np.int64), group_num, padded=True, pinned=done is not None)
        matches = self.match_layers_end(handle)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        b_idx = np.stack([np.concatenate([np.full(len(s), b, np.int64) for b, (s, _) in enumerate(layer)]) for layer in matches])
        q_idx = np.stack([np.concatenate([s for s, _ in layer]) for layer in matches])
        t_idx = np.stack([np.concatenate([t + offs[b] for b, (_, t) in enumerate(layer)]) for layer in matches])
        return np.stack([b_idx, q_idx, t_idx]).astype(np.int64)

    @torch.no_grad()
    def match_layers_end(self, handle):
        from .. import lsap
        payload, NL, B, Q, sizes, group_num = handle
        if payload is None:
            e = np.empty(0, np.int64)
            return [[(e, e) for _ in range(B)] for _ in range(NL)]
        if isinstance(payload[0], str):                                                    # list form of the device solver's pairs (tests, slow path)
            idx = payload[1].cpu().numpy()
            self.check_device_status(block=True)
            g_q, offs = Q // group_num, np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
            per = [group_num * min(g_q, int(n)) for n in sizes]
            first = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
            return [[(idx[1, l, first[b]:first[b + 1]], idx[2, l, first[b]:first[b + 1]] - offs[b]) for b in range(B)] for l in range(NL)]
        blocks, done = payload
        if done is not None:
            done.synchronize()                                                        # the step's one host sync
        host = blocks.numpy()
        if lsap.available():
            return lsap.match_groups(host, np.asarray(sizes, np.int64), group_num, padded=True)
        g_q = Q // group_num
        out = []
        for l in range(NL):
            layer = []
            for b in range(B):
                src, tgt = [], []
                for g in range(group_num):
                    r, c = linear_sum_assignment(host[l, b, g * g_q:(g + 1) * g_q, :sizes[b]])
                    src.append(r + g * g_q)
                    tgt.append(c)
                layer.append((np.concatenate(src), np.concatenate(tgt)))
            out.append(layer)
        return out


def build_matcher(cfg):
    return HungarianMatcher(cost_class=cfg["set_cost_class"], cost_bbox=cfg["set_cost_bbox"],
                            cost_3dcenter=cfg["set_cost_3dcenter"], cost_giou=cfg["set_cost_giou"])
