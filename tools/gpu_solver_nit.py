#!/usr/bin/env python3
"""Round 6: iteration counts and final objective of the bench workload's 3D decode (DLA-34 bs=32: the natural detections of the
synthetic weights), per solver form, and the decode kernel's duration alone (no forward beside it): what the side stream holds
CUs for.  python tools/gpu_solver_nit.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtm3d_amd                                    # noqa: E402
from rtm3d_amd import weights                       # noqa: E402
from rtm3d_amd.model_utils import decode3d_slots    # noqa: E402

dev = torch.device('cuda', 0)
bb = 'DLA-34'
cfg = rtm3d_amd.kitti_config(bb)
m = rtm3d_amd.create_model(cfg).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
B, H, W = 32, 384, 1280
x = weights.synth_images(B, H, W, seed=1234).to(dev)
K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), dtype=torch.float64, device=dev)
det = m.decode2d(m.forward_logits(x))
torch.cuda.synchronize()
n = det.n.cpu().numpy()
print('objects per image: total %d, min %d, max %d' % (n.sum(), n.min(), n.max()))
for form in ('published', 'direct'):
    bx = decode3d_slots(det, K, cfg.DETECTOR.dim_ref, [0, -0.5, 20], form=form)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(5):
        ev[0].record(); decode3d_slots(det, K, cfg.DETECTOR.dim_ref, [0, -0.5, 20], out=bx, form=form); ev[1].record()
        torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    live = (bx.status >= 0).cpu().numpy()
    nit = bx.nit.cpu().numpy()[live]
    fun = bx.fun.cpu().numpy()[live]
    st = bx.status.cpu().numpy()[live]
    q = np.percentile(nit, [50, 90, 99])
    print('%-9s kernel alone %.3f ms (min of 5) | iterations mean %.1f p50 %d p90 %d p99 %d max %d | kept (fun < 0.1) %d of %d | status counts %s'
          % (form, min(ts), nit.mean(), q[0], q[1], q[2], nit.max(), int((fun < 0.1).sum()), len(nit), dict(zip(*np.unique(st, return_counts=True)))))
    # a workgroup of eight consecutive slots lives as long as its slowest object
    slots = bx.nit.cpu().numpy().reshape(-1, 8) * (bx.status.cpu().numpy().reshape(-1, 8) >= 0)
    wg = slots.max(1)
    print('          workgroups with work %d, their max-iteration: mean %.1f max %d; sum over objects %d vs 8 x sum of workgroup maxima %d'
          % (int((wg > 0).sum()), wg[wg > 0].mean(), wg.max(), nit.sum(), 8 * wg.sum()))
