#!/usr/bin/env python3
"""Golden vectors for the 3D-box post-processing step, produced by RUNNING the reference's
`calc_proj_corners` / `create_corners` / `rotation_matrix` (utils/model_utils.py:66-152), the functions its
drawing code feeds the decoded `ParamList` through (utils/visual_utils.py:60-110).  Run only in the build
container:

    cd /tmp && python -B /root/repo/tests/golden/make_golden_project.py

This script contains no reference source; torchvision (dead DeformConv2d import) is stubbed as in make_golden.py.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
tv = types.ModuleType('torchvision'); tvo = types.ModuleType('torchvision.ops'); tvm = types.ModuleType('torchvision.models')
tvo.DeformConv2d = type('DeformConv2d', (), {})
tv.ops, tv.models = tvo, tvm
sys.modules.update({'torchvision': tv, 'torchvision.ops': tvo, 'torchvision.models': tvm})
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
from utils import model_utils as ref_mu  # noqa: E402 (reference)

rng = np.random.Generator(np.random.PCG64(20240607))
N = 48
dims = np.stack([rng.uniform(1.2, 2.2, N), rng.uniform(0.5, 2.0, N), rng.uniform(0.8, 4.5, N)], 1)      # (h, w, l)
locs = np.stack([rng.uniform(-20, 20, N), rng.uniform(0.5, 2.5, N), rng.uniform(4, 60, N)], 1)
rys = rng.uniform(-np.pi, np.pi, N)
# yaw values inside the reference's |sin|,|cos| < 1e-3 snapping window and just outside it
rys[:8] = [0.0, 5e-4, -5e-4, 2e-3, np.pi / 2, np.pi / 2 - 4e-4, -np.pi / 2 + 9e-4, np.pi - 3e-4]
K = np.array([[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]])
proj = np.stack([ref_mu.calc_proj_corners(dims[i], locs[i], rys[i], K) for i in range(N)])              # (N, 9, 2)
corners = np.stack([ref_mu.create_corners(dims[i], locs[i], ref_mu.rotation_matrix(rys[i])) for i in range(N)])   # (N, 3, 9)
np.savez_compressed(os.path.join(HERE, 'project_cases.npz'), dimension=dims, location=locs, Ry=rys, K=K, proj=proj, corners=corners)
print('project_cases.npz', proj.shape, corners.shape)
