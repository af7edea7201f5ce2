"""Checkpoint loading with the reference's semantics (utils/check_point.py:14-92,136-182):
``torch.load(f)['model']`` (or the dict itself), suffix-matched keys, strict load."""
import logging

import torch

logger = logging.getLogger(__name__)


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    """For every model key pick the loaded key that equals it or is its longest '.'-suffix
    (utils/check_point.py:14-63)."""
    loaded_keys = sorted(loaded_state_dict.keys())
    for key in sorted(model_state_dict.keys()):
        best, best_len = None, 0
        for lk in loaded_keys:
            n = len(lk) if key == lk else (len(lk) + 1 if key.endswith('.' + lk) else 0)
            if n > best_len:
                best, best_len = lk, n
        if best is not None:
            model_state_dict[key] = loaded_state_dict[best]


def load_state_dict(model, loaded_state_dict):
    msd = model.state_dict()
    loaded = {k: v for k, v in loaded_state_dict.items() if not k.startswith('model.24.anchors')}
    align_and_update_state_dicts(msd, loaded)
    model.load_state_dict(msd)        # strict


class CheckPointer(object):
    def __init__(self, model, solver=None, save_dir='', save_to_disk=None, logger=None, mode='full', device='cpu'):
        self.model, self.mode, self.save_dir = model, mode, save_dir
        self.device = device

    def load(self, f=None, use_latest=True, load_solver=True):
        if not f:
            logger.info('No checkpoint found. Initializing model from scratch')
            return {}
        ckpt = torch.load(f, map_location='cpu')
        sd = ckpt['model'] if 'model' in ckpt else ckpt
        if self.mode == 'full' and hasattr(sd, 'state_dict'):
            sd = sd.float().state_dict()
        load_state_dict(self.model, sd)
        return ckpt
