"""Shared helpers for the parity tests."""
import os
import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def canon_dets(cls, score, *rest):
    """Canonical order (score desc, flat index is unknown here -> cls, x, y as tie-breakers)."""
    cls = np.asarray(cls); score = np.asarray(score)
    mproj = np.asarray(rest[0])
    order = np.lexsort((mproj[:, 0], mproj[:, 1], cls, -score.astype(np.float64)))
    return [np.asarray(a)[order] for a in (cls, score) + tuple(rest)]


def dets_from_golden(g, prefix, b):
    return [g['%s%s_%d' % (prefix, k, b)] for k in ('cls', 'score', 'mproj', 'verts', 'bbox')]


def to_np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
