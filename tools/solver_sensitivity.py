#!/usr/bin/env python3
"""How far does the REFERENCE's own 3D decode (SciPy L-BFGS-B, utils/model_utils.py:264-312, run here through the CPU
oracle) move when its input vertices move by fp32-level noise?  CPU only; writes the table DESIGN.md section 4 cites.

    python tools/solver_sensitivity.py > profiles/r02_solver_sensitivity.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decode3d_ref                      # noqa: E402
from tests.golden.cases import DIM_REF               # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'planted_small.npz'))
K = g['K']
rng = np.random.default_rng(0)
DRAWS = 8            # independent perturbations per noise level (maxima over all of them)
print('# kept objects of tests/golden/planted_small.npz; vertices perturbed by uniform(-eps, eps) px (%d draws per level), re-solved by the oracle (SciPy)' % DRAWS)
print('# x = [sin, cos, l, h, w, X, Y, Z]; the fp32 verification mode\'s vertex error is <= 3e-5 px on this fixture, the fp16 path\'s 0.023 px')
print('%5s %8s %10s %12s %14s  %s' % ('image', 'eps_px', 'max|dx|', 'median obj', 'nit changed', 'per-parameter max'))
for b in range(2):
    cls, v = g['det_cls_%d' % b], g['det_verts_%d' % b]
    _, raw = decode3d_ref.optim_decode_bbox3d(cls, v, K, DIM_REF, [0, -0.5, 20], return_raw=True)
    kept = raw['kept']
    assert np.abs(raw['x'] - g['d3_raw_x_%d' % b])[kept].max() == 0.0
    for eps in (3e-2, 3e-5, 3e-6, 1e-7):
        ds, nit = [], 0
        for draw in range(DRAWS):
            vp = (v + rng.uniform(-eps, eps, v.shape)).astype(np.float32)
            _, r2 = decode3d_ref.optim_decode_bbox3d(cls, vp, K, DIM_REF, [0, -0.5, 20], return_raw=True)
            ds.append(np.abs(r2['x'] - raw['x'])[kept & r2['kept']])
            nit += int((r2['nit'] != raw['nit'])[kept].sum())
        d = np.concatenate(ds)
        print('%5d %8.0e %10.2e %12.2e %9d / %3d  %s' % (b, eps, d.max(), np.median(d.max(1)), nit, DRAWS * int(kept.sum()),
                                                         ' '.join('%.1e' % t for t in d.max(0))))
