"""Shared helpers for the parity tests."""
import os
import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def solver_tail_stats(x, fun, g):
    """Solver results (x (N,8), fun) against a reference-run solver fixture (raw_x, raw_fun of decode3d_large.npz): keep / reject
    mismatches and, over the objects both keep, the per-box L-inf over [Ry, h, w, l, X, Y, Z] (what optim_decode_bbox3d returns,
    utils/model_utils.py:299-305; the yaw difference taken on the circle): p50 / p99 / max and the share within 1e-4."""
    kept = g['raw_fun'] < 0.1
    k2 = np.asarray(fun) < 0.1

    def box(v):
        return np.concatenate([np.arctan2(v[:, 0], v[:, 1])[:, None], v[:, [3, 4, 2]], v[:, 5:8]], 1)
    both = kept & k2
    d = np.abs(box(np.asarray(x)) - box(g['raw_x']))[both]
    d[:, 0] = np.minimum(d[:, 0], 2 * np.pi - d[:, 0])
    e = d.max(1)
    return {'n': int(len(kept)), 'kept': int(kept.sum()), 'keep_mismatch': int((k2 != kept).sum()), 'p50': float(np.percentile(e, 50)),
            'p99': float(np.percentile(e, 99)), 'max': float(e.max()), 'within_1e-4': float((e <= 1e-4).mean())}


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


def pack_records_reference(n, cls, score, mproj, verts, bbox, topk, boxes=None):
    """Plain-torch definition of the 32-float detection record (rtm3d_amd/distributed.py layout): the checker of the
    HIP kernel rtm3d_pack_records, and the record builder of the CPU (gloo) tests."""
    B = n.shape[0]
    rec = torch.zeros(B, topk, 32, dtype=torch.float32, device=n.device)
    valid = torch.arange(topk, device=n.device)[None, :] < n[:, None].to(torch.int64)
    rec[..., 0] = cls.view(B, topk).to(torch.float32)
    rec[..., 1] = score.view(B, topk)
    rec[..., 2:4] = mproj.view(B, topk, 2)
    rec[..., 4:20] = verts.view(B, topk, 16)
    rec[..., 20:24] = bbox.view(B, topk, 4)
    flag = valid.to(torch.float32)
    if boxes is not None:
        kept = boxes.kept.view(B, topk) & valid
        rec[..., 24:27] = boxes.dimension.view(B, topk, 3).to(torch.float32)
        rec[..., 27:30] = boxes.location.view(B, topk, 3).to(torch.float32)
        rec[..., 30] = boxes.Ry.view(B, topk).to(torch.float32)
        flag = flag + kept.to(torch.float32)
    rec[..., 31] = flag
    return torch.where((flag > 0)[..., None], rec, torch.zeros_like(rec))


# ---- measured-error log: the GPU parity tests record what they measured (not only pass/fail); the file is copied to
# profiles/ by hand after a GPU run (gpurun_out/ is the only directory that travels back from the GPU box)
_MEASURED = {}


def record_measurement(group, key, value):
    import atexit
    import json
    if not _MEASURED:
        def dump():
            out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
            try:
                os.makedirs(out, exist_ok=True)
                with open(os.path.join(out, 'measured_errors.json'), 'w') as f:
                    json.dump(_MEASURED, f, indent=1, sort_keys=True)
            except OSError:
                pass
        atexit.register(dump)
    _MEASURED.setdefault(group, {})[key] = value
