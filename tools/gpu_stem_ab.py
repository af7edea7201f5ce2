#!/usr/bin/env python3
"""Per-op time of the first ops of the bs=32 DLA-34 plan with the two input forms: the caller's fp32 NCHW batch (the fused
stem converts while it stages) against the plan's own fp16 NHWC4 input tensor already filled (rtm3d_forward with d_in = NULL)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtm3d_amd                                    # noqa: E402
from rtm3d_amd import weights, preprocess           # noqa: E402

dev = torch.device('cuda', 0)
bb = 'DLA-34'
cfg = rtm3d_amd.kitti_config(bb)
m = rtm3d_amd.create_model(cfg).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
B, H, W = 32, 384, 1280
x = weights.synth_images(B, H, W, seed=1234).to(dev)
plan = m._plan_for(B, H, W, dev)
outs = [torch.empty(B, c, H // 4, W // 4, dtype=torch.float32, device=dev) for c in (3, 16, 2, 2)]
ptrs = [o.data_ptr() for o in outs]
stream = torch.cuda.current_stream(dev).cuda_stream
gen = torch.Generator().manual_seed(1)
imgs = [torch.randint(0, 256, (360, 1240, 3), generator=gen, dtype=torch.uint8).to(dev) for _ in range(B)]
preprocess.preprocess_batch(imgs, (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, resize_to=1280, model=m)
torch.cuda.synchronize()
for name, d_in in (('fp32 NCHW', x.data_ptr()), ('NHWC4 fp16 preloaded', 0), ('fp32 NCHW', x.data_ptr()), ('NHWC4 fp16 preloaded', 0)):
    plan.forward_timed(stream, d_in, ptrs)
    best = None
    for _ in range(5):
        info = plan.forward_timed(stream, d_in, ptrs)
        if best is None:
            best = [i['ms'] for i in info]
        best = [min(a, i['ms']) for a, i in zip(best, info)]
    print('%-22s total %.3f ms | %s' % (name, sum(best), '  '.join('%s %.3f' % (i['name'][:24], t) for i, t in list(zip(info, best))[:3])))
