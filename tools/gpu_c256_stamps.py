#!/usr/bin/env python3
"""In-kernel timeline of the persistent 256 x 256 convolution (diagnostic build -DC256_STAMPS of the library, see
conv_mfma256.hip): python tools/gpu_c256_stamps.py <stamps librtm3d_hip.so> [batch]
Runs one bs=32 DLA-34 forward; the last launch of conv_mfma256_persistent_kernel is heads.conv_d6, whose stamps (workgroups 0
and 101, waves 0 and 4, second tile of each) are read back and printed as cycles per barrier interval."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtm3d_amd import _lib                          # noqa: E402
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import rtm3d_amd                                    # noqa: E402
from rtm3d_amd import weights                       # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
for _ in range(3):
    info = plan.forward_timed(stream, x.data_ptr(), ptrs)
print('per-op ms:', '  '.join('%s %.3f' % (i['name'][-14:], i['ms']) for i in info if 'heads' in i['name']))
lib = _lib.load()
KT, NS = 20, 20
buf = np.zeros(16384, np.uint32)
_lib.check(lib.rtm3d_ctx_debug_read_words(plan.ctx, 0, 16384, buf.ctypes.data_as(ctypes.c_void_p)), 'debug_read_words')
u64 = buf.view(np.uint64)
NAMES = ['issue+lds', 'vmcnt', 'barrier', 'mfma', 'barrier']
for wg in range(4):
    for half in range(2):
        base = wg * 1024 + half * 512
        if wg >= 2:                     # the halo kernel has LDS for three K-tiles of stamps (K-tiles 4..6)
            st = u64[base:base + 3 * NS].reshape(3, NS).astype(np.int64)
            if int(u64[base + 3 * NS + 16]) == 0:
                print('d1 (halo) slot %d waves %d: no stamps' % (wg % 2, half * 4)); continue
            per = np.zeros((3, NS)); per[:, 1:] = np.diff(st, axis=1); per[1:, 0] = st[1:, 0] - st[:-1, NS - 1]
            mean = per[1:].mean(axis=0)
            print('d1 (halo) workgroup slot %d, wave %d, second tile, K-tiles 5..6: cycles per K-tile %.0f (ideal 2048)' % (wg % 2, half * 4, mean.sum()))
            for ph in range(4):
                print('   phase %d: ' % (ph + 1) + '  '.join('%s %4.0f' % (NAMES[k], mean[ph * 5 + k]) for k in range(5)) + '   | sum %.0f' % mean[ph * 5:ph * 5 + 5].sum())
            continue
        st = u64[base:base + KT * NS].reshape(KT, NS).astype(np.int64)
        ts = u64[base + KT * NS:base + KT * NS + 16].reshape(4, 4).astype(np.int64)
        T = int(u64[base + KT * NS + 16])
        if T == 0:
            print('workgroup slot %d waves %d: no stamps' % (wg, half * 4))
            continue
        T = min(T, KT)
        # interval k: stamp k-1 -> stamp k (k = 0: from the previous K-tile's last stamp)
        per = np.zeros((T, NS))
        per[:, 1:] = np.diff(st[:T], axis=1)
        per[1:, 0] = st[1:T, 0] - st[:T - 1, NS - 1]
        mean = per[2:T - 1].mean(axis=0)
        print('%s workgroup slot %d, wave %d, second tile, K-tiles 2..%d of %d: cycles per K-tile %.0f (ideal 2048)' % ('d6 (generic)' if wg < 2 else 'd1 (halo)', wg % 2, half * 4, T - 2, int(u64[base + KT * NS + 16]), mean.sum()))
        for ph in range(4):
            print('   phase %d: ' % (ph + 1) + '  '.join('%s %4.0f' % (NAMES[k], mean[ph * 5 + k]) for k in range(5)) + '   | sum %.0f' % mean[ph * 5:ph * 5 + 5].sum())
        for ti in range(4):
            if ts[ti, 0]:
                print('  tile %d: K loop %d cycles, epilogue %d, wait for the other half %d%s' %
                      (ti, ts[ti, 1] - ts[ti, 0], ts[ti, 2] - ts[ti, 1], ts[ti, 3] - ts[ti, 2],
                       ', gap to next tile %d' % (ts[ti + 1, 0] - ts[ti, 3]) if ti < 3 and ts[ti + 1, 0] else ''))
