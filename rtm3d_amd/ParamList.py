"""Dict-of-arrays result container returned by the 3D decode (own implementation of the
interface of utils/ParamList.py:16-144 that the hot path and its consumers use)."""
import copy

import numpy as np
import torch


class ParamList(object):
    def __init__(self, image_size, is_training=True):
        self.size = image_size
        self.is_training = is_training
        self.extra_fields = {}

    def add_field(self, field, field_data, to_tensor=False):
        if to_tensor and not isinstance(field_data, torch.Tensor):
            field_data = torch.as_tensor(field_data)
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def update_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def to(self, device):
        for k, v in self.extra_fields.items():
            if hasattr(v, 'to'):
                self.extra_fields[k] = v.to(device)
        return self

    def numpy(self):
        c = ParamList(self.size, self.is_training)
        for k, v in self.extra_fields.items():
            if isinstance(v, torch.Tensor):
                c.extra_fields[k] = v.detach().cpu().numpy()
            elif isinstance(v, np.ndarray):
                c.extra_fields[k] = np.copy(v)
            else:
                c.extra_fields[k] = copy.deepcopy(v)
        return c

    def copy_field(self, other, fields):
        for f in fields:
            if other.has_field(f):
                self.add_field(f, copy.deepcopy(other.get_field(f)))

    def __len__(self):
        m = self.extra_fields.get('mask')
        return int(np.count_nonzero(np.asarray(m))) if (self.is_training and m is not None) else 0

    def __repr__(self):
        w, h = (self.size if self.size is not None else (None, None))
        return 'ParamList(regress_number=%d, image_width=%s, image_height=%s)' % (len(self), w, h)
