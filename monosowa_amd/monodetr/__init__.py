"""MonoDETR on MI355X: ``build(cfg['model']) -> (model, criterion)`` like the reference's
lib/models/monodetr/__init__.py:4 + monodetr.py:1293-1360."""
import torch

from .backbone import build_backbone
from .criterion import SetCriterion
from .depth_predictor import DepthPredictor
from .depthaware_transformer import build_depthaware_transformer
from .matcher import build_matcher
from .monodetr import MonoDETR


def build_weight_dict(cfg):
    wd = {"loss_ce": cfg["cls_loss_coef"], "loss_bbox": cfg["bbox_loss_coef"], "loss_giou": cfg["giou_loss_coef"],
          "loss_dim": cfg["dim_loss_coef"], "loss_angle": cfg["angle_loss_coef"], "loss_depth": cfg["depth_loss_coef"],
          "loss_center": cfg["3dcenter_loss_coef"], "loss_depth_map": cfg["depth_map_loss_coef"],
          "loss_tfl": cfg["tfl_loss_coef"], "loss_mask": cfg["mask_loss_coef"]}
    if cfg["aux_loss"]:
        aux = {}
        for i in range(cfg["dec_layers"] - 1):
            aux.update({k + f"_{i}": v for k, v in wd.items()})
        aux.update({k + "_enc": v for k, v in wd.items()})
        wd.update(aux)
    return wd


def build(cfg):
    if cfg.get("use_dn", False):
        raise NotImplementedError("use_dn (denoising queries) is off in every shipped config")
    backbone = build_backbone(cfg)
    transformer = build_depthaware_transformer(cfg)
    depth_predictor = DepthPredictor(cfg)
    model = MonoDETR(backbone, transformer, depth_predictor, num_classes=cfg["num_classes"],
                     num_queries=cfg["num_queries"], aux_loss=cfg["aux_loss"],
                     num_feature_levels=cfg["num_feature_levels"], with_box_refine=cfg["with_box_refine"],
                     two_stage=cfg["two_stage"], init_box=cfg["init_box"], use_dab=cfg["use_dab"],
                     two_stage_dino=cfg["two_stage_dino"])
    matcher = build_matcher(cfg)
    losses = ["labels", "boxes", "cardinality", "depths", "dims", "angles", "center", "depth_map", "tfl"]
    criterion = SetCriterion(cfg["num_classes"], matcher=matcher, weight_dict=build_weight_dict(cfg),
                             focal_alpha=cfg["focal_alpha"], losses=losses, cfg=cfg,
                             depth_map_size=tuple(cfg.get("depth_map_size", (80, 24))))
    device = torch.device(cfg["device"])
    if device.type == "cuda" and not torch.cuda.is_available():
        device = torch.device("cpu")     # CPU plumbing config (BASELINE configs[0])
    criterion.to(device)
    return model, criterion
