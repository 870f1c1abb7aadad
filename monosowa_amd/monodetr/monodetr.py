"""MonoDETR detector: backbone -> input projections -> depth predictor -> depth-aware transformer
-> per-decoder-layer heads.

Reference: lib/models/monodetr/monodetr.py -- ``MonoDETR`` :34-299, ``MLP`` :1278-1290.
Same constructor arguments, parameter names, forward signature
``model(images, calibs, targets, img_sizes, dn_args=None)`` and output dict keys
(pred_logits, pred_boxes, pred_3d_dim, pred_depth, pred_angle, pred_depth_map_logits, aux_outputs).
Only the branch the shipped configs run is restated (two_stage / use_dab / two_stage_dino /
use_dn are all False).
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from ..pointwise import conv_group_norm, head_tail, head_tail_supported
from .depthaware_transformer import MLP, merged_first_layers
from .misc import NestedTensor, inverse_sigmoid


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


MERGE_HEADS = True
FUSED_HEAD_TAIL = True     # box sigmoid + regressed / geometric / depth-map depth average per level as one kernel each way
REUSE_BBOX_RAW = True
USE_LAYER_TENSORS = True   # heads read the decoder's per-layer tensors instead of selects of its stacked outputs


def tmp_supported(ref):
    """reference boxes the fused head tail can take itself: detached float32 CUDA [B, Q, 2 | 6]"""
    return ref.is_cuda and ref.dtype == torch.float32 and not ref.requires_grad and ref.dim() == 3 and ref.shape[-1] in (2, 6)


class MonoDETR(nn.Module):
    def __init__(self, backbone, depthaware_transformer, depth_predictor, num_classes, num_queries,
                 num_feature_levels, aux_loss=True, with_box_refine=False, two_stage=False, init_box=False,
                 use_dab=False, group_num=11, two_stage_dino=False):
        super().__init__()
        if two_stage or use_dab or two_stage_dino:
            raise NotImplementedError("two_stage / use_dab / two_stage_dino are off in every shipped config")
        self.num_queries = num_queries
        self.group_num = group_num
        self.depthaware_transformer = depthaware_transformer
        self.depth_predictor = depth_predictor
        hidden_dim = depthaware_transformer.d_model
        self.hidden_dim = hidden_dim
        self.num_feature_levels = num_feature_levels
        self.two_stage_dino = False
        self.label_enc = nn.Embedding(num_classes + 1, hidden_dim - 1)   # DN indicator table, unused here

        # prediction heads
        class_embed = nn.Linear(hidden_dim, num_classes)
        prior_prob = 0.01
        class_embed.bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        bbox_embed = MLP(hidden_dim, hidden_dim, 6, 3)       # (cx, cy, l, r, t, b)
        dim_embed_3d = MLP(hidden_dim, hidden_dim, 3, 2)
        angle_embed = MLP(hidden_dim, hidden_dim, 24, 2)     # 12 heading bins + 12 residuals
        depth_embed = MLP(hidden_dim, hidden_dim, 2, 2)      # depth and log-variance
        self.use_dab = False
        if init_box:
            nn.init.constant_(bbox_embed.layers[-1].weight.data, 0)
            nn.init.constant_(bbox_embed.layers[-1].bias.data, 0)

        self.query_embed = nn.Embedding(num_queries * group_num, hidden_dim * 2)

        if num_feature_levels > 1:
            projs, in_ch = [], None
            for in_ch in backbone.num_channels:
                projs.append(nn.Sequential(nn.Conv2d(in_ch, hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim)))
            for _ in range(num_feature_levels - len(backbone.num_channels)):
                projs.append(nn.Sequential(nn.Conv2d(in_ch, hidden_dim, kernel_size=3, stride=2, padding=1),
                                           nn.GroupNorm(32, hidden_dim)))
                in_ch = hidden_dim
            self.input_proj = nn.ModuleList(projs)
        else:
            self.input_proj = nn.ModuleList([nn.Sequential(
                nn.Conv2d(backbone.num_channels[0], hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim))])

        self.backbone = backbone
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine
        self.two_stage = False
        self.num_classes = num_classes

        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

        num_pred = depthaware_transformer.decoder.num_layers
        if with_box_refine:
            self.class_embed = _clones(class_embed, num_pred)
            self.bbox_embed = _clones(bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.depthaware_transformer.decoder.bbox_embed = self.bbox_embed      # shared with the decoder
            self.dim_embed_3d = _clones(dim_embed_3d, num_pred)
            self.depthaware_transformer.decoder.dim_embed = self.dim_embed_3d
            self.angle_embed = _clones(angle_embed, num_pred)
            self.depth_embed = _clones(depth_embed, num_pred)
        else:
            nn.init.constant_(bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([bbox_embed for _ in range(num_pred)])
            self.dim_embed_3d = nn.ModuleList([dim_embed_3d for _ in range(num_pred)])
            self.angle_embed = nn.ModuleList([angle_embed for _ in range(num_pred)])
            self.depth_embed = nn.ModuleList([depth_embed for _ in range(num_pred)])
            self.depthaware_transformer.decoder.bbox_embed = None
            # NB: the reference leaves decoder.dim_embed = None here and then fails at :623; kept as is.

    # parameters the shipped forward never touches (no gradient): DDP must not wait for them
    UNUSED_PARAMETER_PREFIXES = ("label_enc.", "depthaware_transformer.decoder.query_scale.",
                                 "depthaware_transformer.decoder.ref_point_head.")
    UNUSED_PARAMETER_SUFFIXES = (".sa_v_proj.weight", ".sa_v_proj.bias")

    def unused_parameter_names(self):
        return [n for n, _ in self.named_parameters()
                if n.startswith(self.UNUSED_PARAMETER_PREFIXES) or n.endswith(self.UNUSED_PARAMETER_SUFFIXES)]

    def _project(self, l, x):
        """input_proj[l] = Conv2d + GroupNorm(32, hidden_dim) (monodetr.py:68-88), the norm through the NHWC kernel."""
        conv, gn = self.input_proj[l]
        return conv_group_norm(x, conv, gn)

    def project_features(self, features, pos):
        srcs, masks = [], []
        all_valid = all(getattr(f, "all_valid", False) for f in features)
        for l, feat in enumerate(features):
            src, mask = feat.decompose()
            assert mask is not None
            srcs.append(self._project(l, src))
            masks.append(mask)
        for l in range(len(srcs), self.num_feature_levels):   # extra stride-2 levels from C5
            last = features[-1].tensors
            src = self._project(l, getattr(last, "fork_twin", last) if l == len(features) else srcs[-1])      # (c5's second handle)
            mask = torch.zeros(src.shape[0], src.shape[2], src.shape[3], dtype=torch.bool, device=src.device)
            pos.append(self.backbone[1](NestedTensor(src, mask, all_valid=True)).to(src.dtype))
            srcs.append(src)
            masks.append(mask)
        self._all_valid = all_valid          # the extra levels are built with all-False masks above
        return srcs, masks, pos

    def forward(self, images, calibs, targets, img_sizes, dn_args=None):
        """images [B,3,H,W]; calibs [B,3,4] (only fu = calibs[:,0,0] is read); img_sizes [B,2]
        (only the height column is read); targets / dn_args unused on the shipped path."""
        features, pos = self.backbone(images)
        srcs, masks, pos = self.project_features(features, pos)

        query_embeds = self.query_embed.weight if self.training else self.query_embed.weight[:self.num_queries]

        pred_depth_map_logits, depth_pos_embed, weighted_depth, depth_pos_embed_ip = \
            self.depth_predictor(srcs, masks[1], pos[1], all_valid=self._all_valid)

        hs, init_reference, inter_references, inter_references_dim, _, _ = self.depthaware_transformer(
            srcs, masks, pos, query_embeds, depth_pos_embed, depth_pos_embed_ip, all_valid=self._all_valid)

        coords, classes, dims3d, depths, angles = [], [], [], [], []
        fu = calibs[:, 0, 0].unsqueeze(1)
        img_h = img_sizes[:, 1:2]
        img_h_f = img_h.to(torch.float32) if FUSED_HEAD_TAIL else img_h     # loaders hand int32 sizes; the products below promote them
        # the decoder's per-layer tensors, not selects of its stacks (same values; see DepthAwareDecoder.forward)
        dec = self.depthaware_transformer.decoder
        hs_l = getattr(dec, "inter_outputs", None)
        hs_l = hs_l if hs_l is not None and len(hs_l) == hs.shape[0] and USE_LAYER_TENSORS else hs.unbind(0)
        dims_l = getattr(dec, "inter_dims", None)
        dims_l = dims_l if dims_l is not None and len(dims_l) == hs.shape[0] and USE_LAYER_TENSORS else inter_references_dim.unbind(0)
        for lvl in range(hs.shape[0]):
            ref_raw = init_reference if lvl == 0 else inter_references[lvl - 1]
            # detached reference boxes (every level but the first, whose points come from the query embedding): the fused head
            # tail adds inverse_sigmoid(reference) to the box logits itself
            ref_in_kernel = FUSED_HEAD_TAIL and MERGE_HEADS and tmp_supported(ref_raw)
            reference = None if ref_in_kernel else inverse_sigmoid(ref_raw)
            # the decoder evaluated bbox_embed[lvl] on the same hs[lvl] for its reference refinement (whose result it
            # detaches); the reference evaluates it a second time here (monodetr.py:222) -- same values, so reuse the
            # tensor (its graph carries the gradient the recomputation would have produced)
            raw = getattr(self.depthaware_transformer.decoder, "bbox_raw", None)
            tmp = raw[lvl] if raw and len(raw) == hs.shape[0] and REUSE_BBOX_RAW else self.bbox_embed[lvl](hs_l[lvl])
            if reference is None:
                pass
            elif reference.shape[-1] == 6:
                tmp = tmp + reference
            else:
                assert reference.shape[-1] == 2
                t_xy, t_rest = tmp.split([2, tmp.shape[-1] - 2], -1)
                tmp = torch.cat([t_xy + reference, t_rest], -1)
            if MERGE_HEADS:
                # class / depth / angle heads read the same hs[lvl]: first layers as one GEMM
                cls, depth_reg, angle = merged_first_layers(hs_l[lvl], [self.class_embed[lvl], self.depth_embed[lvl], self.angle_embed[lvl]])
            else:
                cls, depth_reg, angle = self.class_embed[lvl](hs_l[lvl]), None, None
            classes.append(cls)
            size3d = dims_l[lvl]
            dims3d.append(size3d)
            if FUSED_HEAD_TAIL and depth_reg is not None and head_tail_supported(tmp, size3d, depth_reg, weighted_depth, fu, img_h_f):
                # sigmoid of the box logits + the three-way depth average below: one HIP kernel each way (csrc/head_tail.hip)
                outputs_coord, depth_ave = head_tail(tmp, size3d, depth_reg, weighted_depth, fu, img_h_f,
                                                     ref=ref_raw if reference is None else None)
                coords.append(outputs_coord)
                depths.append(depth_ave)
                angles.append(angle)
                continue
            if reference is None:                                           # (the fused path was planned but does not apply)
                inv = inverse_sigmoid(ref_raw)
                tmp = tmp + inv if inv.shape[-1] == 6 else torch.cat([tmp[..., :2] + inv, tmp[..., 2:]], -1)
            outputs_coord = tmp.sigmoid()                                   # 3D-centre projection + l,r,t,b
            coords.append(outputs_coord)

            # geometric depth from the 3D height and the predicted 2D box height (monodetr.py:246-248)
            box2d_height = torch.clamp((outputs_coord[:, :, 4] + outputs_coord[:, :, 5]) * img_h, min=1.0)
            depth_geo = size3d[:, :, 0] / box2d_height * fu
            if depth_reg is None:
                depth_reg = self.depth_embed[lvl](hs_l[lvl])
            # depth read from the predicted depth map at the projected 3D centre (:254-259)
            centre = ((outputs_coord[..., :2] - 0.5) * 2).unsqueeze(2).detach()
            depth_map = F.grid_sample(weighted_depth.unsqueeze(1), centre, mode="bilinear",
                                      align_corners=True).squeeze(1)
            depth_ave = torch.cat([((1. / (depth_reg[:, :, 0:1].sigmoid() + 1e-6) - 1.)
                                    + depth_geo.unsqueeze(-1) + depth_map) / 3,
                                   depth_reg[:, :, 1:2]], -1)
            depths.append(depth_ave)
            angles.append(angle if angle is not None else self.angle_embed[lvl](hs_l[lvl]))

        out = {"pred_logits": classes[-1], "pred_boxes": coords[-1], "pred_3d_dim": dims3d[-1],
               "pred_depth": depths[-1], "pred_angle": angles[-1],
               "pred_depth_map_logits": pred_depth_map_logits}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "pred_3d_dim": c, "pred_angle": d, "pred_depth": e}
                                  for a, b, c, d, e in zip(classes[:-1], coords[:-1], dims3d[:-1], angles[:-1], depths[:-1])]
        return out
