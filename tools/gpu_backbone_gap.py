#!/usr/bin/env python3
"""Round 6: where the DLA-34 backbone's time goes, layer by layer.  Per-op hipEvent times (min over 8 replays of
rtm3d_forward_timed) of the bs=32 plan against the SAME kernels at bs=128 (four times the work items: many rounds, the launch's
fixed cost and the last round's quantisation amortised four times further):
   ms32            the layer at bs=32
   ms128/4         what 32 images cost inside a bs=128 launch
   fixed+quant     ms32 - ms128/4: per-launch fixed cost (launch gap, prologue, epilogue drain) + the bs=32 launch's partial last round
   at_1300         the layer's FLOPs at 1300 TFLOP/s (the head convs' rate on this data: the power-limited MFMA rate of the chip)
   kernel_gap      ms128/4 - max(at_1300, bytes at 5 TB/s): what the kernel's own K loop / memory path leaves against that
usage: python tools/gpu_backbone_gap.py > profiles/r06_backbone_gap.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtm3d_amd                                    # noqa: E402
from rtm3d_amd import weights                       # noqa: E402

dev = torch.device('cuda', 0)
bb = 'DLA-34'
m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
H, W = 384, 1280


def per_op(B):
    x = weights.synth_images(B, H, W, seed=1234).to(dev)
    plan = m._plan_for(B, H, W, dev)
    outs = [torch.empty(B, c, H // 4, W // 4, dtype=torch.float32, device=dev) for c in (3, 16, 2, 2)]
    ptrs = [o.data_ptr() for o in outs]
    stream = torch.cuda.current_stream(dev).cuda_stream
    plan.forward_timed(stream, x.data_ptr(), ptrs)
    best = None
    for _ in range(8):
        info = plan.forward_timed(stream, x.data_ptr(), ptrs)
        ms = [i['ms'] for i in info]
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    return info, best


i32, t32 = per_op(32)
i128, t128 = per_op(128)
assert [i['name'] for i in i32] == [i['name'] for i in i128], 'the two plans record different ops'
print('%-44s %-28s %7s %8s %11s %8s %10s %7s' % ('op (bs=32 DLA-34 384x1280)', 'kernel', 'ms32', 'ms128/4', 'fixed+quant', 'at_1300', 'kernel_gap', 'TFLOP/s'))
tot = [0.0] * 5
for a, ta, b, tb in zip(i32, t32, i128, t128):
    if not a['name'].startswith('backbone'):
        continue
    k = a['kernel'] if a['kernel'] == b['kernel'] else a['kernel'] + ' | ' + b['kernel']
    at = a['flops'] / 1300e12 * 1e3
    mem = a['bytes'] / 5e12 * 1e3
    many = tb / 4
    row = (ta, many, ta - many, at, many - max(at, mem))
    tot = [x + y for x, y in zip(tot, row)]
    print('%-44s %-28s %7.3f %8.3f %11.3f %8.3f %10.3f %7.0f' % (a['name'][:44], k[:28], row[0], row[1], row[2], row[3], row[4], a['flops'] / ta / 1e9 if ta else 0))
fl = sum(a['flops'] for a in i32 if a['name'].startswith('backbone'))
print('%-44s %-28s %7.3f %8.3f %11.3f %8.3f %10.3f' % ('backbone total', '', *tot))
frac = lambda ms: fl / (ms * 1e-3) / 2.5e15
print('backbone %.1f GFLOP per 32 images: %.3f of 2.5 PFLOP/s at bs=32 (sum of the ops\' own times), %.3f at the bs=128 per-image rate, %.3f with every '
      'layer at 1300 TFLOP/s or 5 TB/s' % (fl / 1e9, frac(tot[0]), frac(tot[1]), frac(tot[1] - tot[4])))
