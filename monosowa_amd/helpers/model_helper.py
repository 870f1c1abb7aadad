"""``build_model(cfg['model']) -> (model, criterion)`` (reference: lib/helpers/model_helper.py:4-5)."""
from ..monodetr import build


def build_model(cfg):
    return build(cfg)
