"""Diagnostic: per-phase cycle counts of the wave-cooperative L-BFGS-B solver on the 64 golden objects.
Needs a library whose decode3d.hip was compiled with -DLBW_PROF=k (tools/prof_lbw.sh builds them into rtm3d_amd/_C/prof<k>):
the solver then writes eight of its phase cycle sums (s_memtime) where the solution normally goes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtm3d_amd import _lib, model_utils  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])      # the diagnostic build, not the in-tree library
view = int(sys.argv[2]) if len(sys.argv) > 2 else 1
form = sys.argv[3] if len(sys.argv) > 3 else 'published'
NAMES = {'direct': {1: ['direction (two-loop)', '-', 'line search', 'matupd', 'ls: f + g', 'ls: dcsrch', '-', 'total']},
         # lbfgsb_wave_pub.h: LBW_PROF = 1 / 2 / 3 selects which eight of its 24 counters come back
         'published': {1: ['formk', 'subsm', 'line search', 'matupd', 'formt', '-', '-', 'total'],
                       2: ['formk: shift', 'formk: new row', 'formk: WN from WN1', 'formk: potrf2 (1,1)+T', 'formk: rhs solves', 'formk: (2,2) update', 'formk: potrf (2,2)', '-'],
                       3: ['subsm: trsv_ut', 'subsm: trsv_un', '-', '-', '-', '-', '-', '-']}}[form][view]
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'decode3d_cases.npz'))
x, fun, nit, st = model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], form=form)
ok = st == 0
if view == 1:
    print('objects', len(nit), 'converged', int(ok.sum()), 'iterations mean %.1f max %d' % (nit.mean(), nit.max()))
for k, name in enumerate(NAMES):
    if name != '-':
        v = x[ok, k]
        print('%-14s cycles/object %9.0f   per iteration %7.0f' % (name, v.mean(), (v / np.maximum(nit[ok], 1)).mean()))
