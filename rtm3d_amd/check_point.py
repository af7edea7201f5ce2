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
        # weights_only=True, stated: tensors and plain containers only, nothing from the file is executed (N ranks of a multi-GPU
        # start all read it).  The reference's mode='full' files pickle the whole nn.Module (utils/check_point.py:120-122,
        # torch.save({'model': model})): unpickling one would run code from the file, so it is refused with the way out.
        try:
            ckpt = torch.load(f, map_location='cpu', weights_only=True)
        except Exception as e:
            raise RuntimeError("CheckPointer.load: %r is not a tensors-only checkpoint (%s: %s).  A mode='full' file of the reference "
                               "holds a pickled module; re-save it with the reference as {'model': model.state_dict()} "
                               "(its mode='state-dict', utils/check_point.py:123-124) and load that." % (f, type(e).__name__, e)) from e
        sd = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
        if not isinstance(sd, dict) or not all(torch.is_tensor(v) for v in sd.values()):
            raise RuntimeError("CheckPointer.load: %r does not hold a state dict of tensors under 'model'" % (f,))
        load_state_dict(self.model, sd)
        return ckpt
