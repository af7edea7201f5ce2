#!/usr/bin/env python3
"""Per-op time (min over replays of rtm3d_forward_timed) of the bs=32 DLA-34 ops whose name matches a pattern, for one build of
the library:  python tools/gpu_variants.py <librtm3d_hip.so> <regex> [batch]   (used by tools/gpu_variants.sh: several builds
interleaved on ONE box - boxes differ by several per cent)."""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtm3d_amd import _lib                          # noqa: E402
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import rtm3d_amd                                    # noqa: E402
from rtm3d_amd import weights                       # noqa: E402

pat = re.compile(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device('cuda', 0)
bb = 'DLA-34'
m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
H, W = 384, 1280
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
sel = [(i['name'], t) for i, t in zip(info, best) if pat.search(i['name'])]
bk = sum(t for i, t in zip(info, best) if i['name'].startswith('backbone'))
print('%-28s backbone %.3f total %.3f | %s' % (os.path.basename(os.path.dirname(sys.argv[1])), bk, sum(best),
                                              '  '.join('%s %.4f' % (n.replace('backbone.', '')[:22], t) for n, t in sel)))
