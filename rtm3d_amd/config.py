"""Configuration tree with the reference's key names (models/configs/detault.py:3-88) and a yaml
merge that accepts the reference's own files (models/configs/rtm3d_*_kitti.yaml).  fvcore/yacs are
not needed: a small attribute dict is enough for the keys the hot path reads
(MODEL.*, DATASET.OBJs, DETECTOR.*, DEVICE)."""
import ast
import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def update(self, *a, **kw):
        dict.update(self, *a, **kw)

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                node = self.get(k)
                if not isinstance(node, CfgNode):
                    node = CfgNode()
                    self[k] = node
                node.merge_from_dict(v)
            else:
                if isinstance(v, str) and v[:1] in '([':      # "(1280, 1280)" -> tuple, like fvcore's eval
                    try:
                        v = ast.literal_eval(v)
                    except (ValueError, SyntaxError):
                        pass
                self[k] = v
        return self

    def merge_from_file(self, path):
        with open(path, 'r') as f:
            return self.merge_from_dict(yaml.safe_load(f) or {})


def _defaults():
    c = CfgNode()
    c.INPUT_SIZE = (640, 640)
    c.BATCH_SIZE = 32
    c.DEVICE = 'cuda'
    c.DATASET = CfgNode(OBJs=['Car', 'Pedestrian', 'Cyclist'], MEAN=[0.485, 0.456, 0.406], STD=[0.229, 0.224, 0.225],
                        VERTEX_OFFSET_INFER=[0.75, 0.57])
    c.MODEL = CfgNode(BACKBONE='DLA-34', DOWN_SAMPLE=4., OUT_CHANNELS=256,
                      KFNs=['level2', 'level3', 'level4', 'level5'], HEADER_NUM_CONV=2,
                      HEAD_VARIANT='rtm3d')   # 'smoke': head-table variant, see rtm3d_amd/weights.py
    c.DETECTOR = CfgNode(CHECKPOINT='./weights/DLA-34/model_0000004.pt', SCORE_THRESH=0.5, TOPK_CANDIDATES=30,
                         NMS_THRESH_TEST=0.5)
    return c


CONFIGS = _defaults()

# the two shipped model configs of the reference (models/configs/rtm3d_{dla34,resnet18}_kitti.yaml)
_DIM_REF = [[1.52607842, 1.62858147, 3.88396124], [1.76067766, 0.6602296, 0.84220464],
            [1.73712792, 0.59677122, 1.76338868]]


def kitti_config(backbone='DLA-34'):
    c = CONFIGS.clone()
    c.INPUT_SIZE = (1280, 1280)
    c.MODEL.BACKBONE = backbone
    if 'RESNET' in backbone:
        c.MODEL.KFNs = ['layer1', 'layer2', 'layer3', 'layer4']
    c.DETECTOR.SCORE_THRESH = 0.4
    c.DETECTOR.TOPK_CANDIDATES = 100
    c.DETECTOR.dim_ref = copy.deepcopy(_DIM_REF)
    return c
