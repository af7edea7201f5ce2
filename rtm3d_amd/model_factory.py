"""create_model(cfg) with the reference signature (models/model_factory.py:23-37)."""
from .model import Model
from .weights import parse_backbone


def create_model(configs):
    """Create the MI355X model for configs.MODEL.BACKBONE ('DLA-34' | 'RESNET-18' | 'RESNET-34')."""
    parse_backbone(configs.MODEL.BACKBONE)     # raises AssertionError('Undefined model backbone') like the reference
    return Model(configs, configs.MODEL.BACKBONE)
