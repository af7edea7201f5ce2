#!/usr/bin/env python3
"""Race detector for the counted-wait kernels (round 5): the bs=32 DLA-34 forward replayed N times, every logit map compared bit for
bit with the first replay's - alone, and with the 3D decode of the previous batch running beside it (Detect3DPipeline), which shifts
every kernel's timing.  A half-tile read before it is retired shows up as a handful of different logits in some replay.
    python tools/gpu_repeat_identity.py [replays] > gpurun_out/repeat_identity.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtm3d_amd
from rtm3d_amd import weights
from rtm3d_amd.pipeline import Detect3DPipeline

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda', 0)
bad = 0
for bb, B, H, W in (('DLA-34', 32, 384, 1280), ('RESNET-18', 8, 384, 1280), ('DLA-34', 3, 96, 320)):
    m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
    m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-4.0))
    x = weights.synth_images(B, H, W, seed=7).to(dev)
    first = [t.clone() for t in m.forward_logits(x)]
    torch.cuda.synchronize()
    diff = 0
    for i in range(N):
        out = m.forward_logits(x, out='reuse')
        diff += sum(int(not torch.equal(a, b)) for a, b in zip(first, out))
    torch.cuda.synchronize()
    print('%-10s bs=%-2d %dx%d  forward alone : %d replays, %d maps differ from the first replay' % (bb, B, H, W, N, diff))
    bad += diff
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), dtype=torch.float64, device=dev)
    pipe = Detect3DPipeline(m, B, dev, gather=False)
    ref = None
    diff = 0
    for i in range(N):
        k = pipe.submit(x, K)
        if i >= 1:
            r = pipe.results(k - 1, copy=True)
            if ref is None:
                ref = r
            else:
                diff += int(not torch.equal(ref, r))
    pipe.drain()
    print('%-10s bs=%-2d %dx%d  pipelined     : %d steps, %d record sets differ from the first' % (bb, B, H, W, N, diff))
    bad += diff
    del pipe, m
print('TOTAL mismatches: %d' % bad)
sys.exit(1 if bad else 0)
