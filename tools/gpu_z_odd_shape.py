#!/usr/bin/env python3
"""z error of the fp16 path and of the fp32 verification mode against the oracle on an odd full-size shape (is the larger z error at
352 x 1216 storage rounding through a peaked spatial softmax, or a wiring bug on shapes the halo kernels do not take?)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtm3d_amd
from rtm3d_amd import weights
from oracle import rtm3d_ref

dev = torch.device('cuda', 0)
for bb, (B, H, W) in [('DLA-34', (2, 352, 1216)), ('DLA-34', (1, 416, 1280)), ('RESNET-18', (1, 352, 1216))]:
    sd = weights.synth_state_dict(bb, 11, 'trained', heat_bias=-3.0)
    x = weights.synth_images(B, H, W, seed=77)
    m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
    m.load_state_dict(sd)
    m.forward_logits(x.to(dev))
    torch.set_num_threads(16)
    _, lref, st = rtm3d_ref.model_forward(x, sd, bb, return_stages=True)
    plan = m._plan_for(B, H, W, dev)
    z16 = plan.download(plan.plan.named['z'])
    m.forward_logits_fp32(x.to(dev))
    z32 = m._verify[1].fetch('z').cpu().numpy()
    zr = st['z'].numpy()
    sc = max(1.0, np.abs(zr).max())
    for nm, z in (('fp16', z16), ('fp32', z32)):
        d = np.abs(z - zr)
        i = np.unravel_index(d.argmax(), d.shape)
        print(bb, (B, H, W), nm, 'rel err %.3e  scale %.2f  worst at %s: got %.4f ref %.4f; p99.9 of |err|/scale %.2e' % (d.max() / sc, sc, i, z[i], zr[i], np.percentile(d, 99.9) / sc))
    # how peaked is the softmax at the worst channel? share of the top pixel of each fusion operand is not stored; report z0-less quantity instead
