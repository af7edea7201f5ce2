#!/usr/bin/env python3
"""where does the end-to-end box error of the fp16 product path come from?

The benchmark's planted images (bench.py `parity`: 8 benchmark images, 16 exact cuboid projections planted per image on the
oracle's logits and carried through the device network additively) are decoded from logits produced by MIXED-PRECISION replays
of the recorded plan: the fp32 verification executor (rtm3d_amd/verify.py) with chosen stages emulated in the product's storage
precision (weights and written activations rounded to fp16, accumulation unchanged).  The all-fp16 emulation is checked against
the real product path first.  Per variant: vertex error of the planted detections (px), per-box L-inf over [Ry,h,w,l,X,Y,Z] on
the boxes both sides keep, share within north_star's 1e-4.  Writes gpurun_out/r04_error_apportioning.txt."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtm3d_amd                                            # noqa: E402
from rtm3d_amd import weights                                # noqa: E402
from rtm3d_amd.model_utils import decode3d_slots             # noqa: E402
from oracle import rtm3d_ref, decode3d_ref                    # noqa: E402  (the checker)
from tests.golden.cases import plant_cuboids                  # noqa: E402

dev = torch.device('cuda', 0)
bb, k, planted = 'DLA-34', 8, 16
cfg = rtm3d_amd.kitti_config(bb)
th, tk, dim_ref = float(cfg.DETECTOR.SCORE_THRESH), int(cfg.DETECTOR.TOPK_CANDIDATES), cfg.DETECTOR.dim_ref
sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0)           # bench.py's workload
m = rtm3d_amd.create_model(cfg).to(dev).eval()
m.load_state_dict(sd)
x = weights.synth_images(k, 384, 1280, seed=1234)
K = weights.synth_intrinsics()
Kd = torch.as_tensor(np.tile(K, (k, 1)), dtype=torch.float64, device=dev)
torch.set_num_threads(min(16, os.cpu_count() or 1))
dets_ref, logits_ref = rtm3d_ref.model_forward(x, sd, bb, th, tk)
lg = [l.numpy().copy() for l in logits_ref]
truth = plant_cuboids(lg[0], lg[1:], K, planted, np.random.Generator(np.random.PCG64(2)))
dets_p = rtm3d_ref.inference([torch.from_numpy(a) for a in lg], th, tk, 4.0)
raws = [None if dets_p[0][b] is None else decode3d_ref.optim_decode_bbox3d(dets_p[0][b].numpy(), dets_p[3][b].numpy(), K, dim_ref, [0, -0.5, 20], return_raw=True)[1]
        for b in range(k)]
only = set((b, c, xx, yy) for b in range(k) for (c, yy, xx, _, _, _) in truth[b])


def box_params(xs):
    return np.concatenate([np.arctan2(xs[:, 0:1], xs[:, 1:2]), xs[:, 3:5], xs[:, 2:3], xs[:, 5:8]], 1)


def stats(logits_dev):
    """planted-additive logits -> product decode kernels -> errors on the planted objects"""
    lgd = [l[:k] + torch.from_numpy(p_ - n_.numpy()).to(dev) for l, p_, n_ in zip(logits_dev, lg, logits_ref)]
    det = m.decode2d(lgd)
    boxes = decode3d_slots(det, Kd, dim_ref, [0, -0.5, 20])
    torch.cuda.synchronize()
    n_dev = det.n.cpu().numpy()
    xs, fs = boxes.x.cpu().numpy(), boxes.fun.cpu().numpy()
    vd, boxd, missed, ref_n = [], [], 0, 0
    for b in range(k):
        nd = int(n_dev[b]); sl = slice(b * tk, b * tk + nd)
        cl, mp, vv = (t[sl].cpu().numpy() for t in (det.cls, det.mproj, det.verts))
        cells = {(int(c), int(q[0] // 4), int(q[1] // 4)): j for j, (c, q) in enumerate(zip(cl, mp))}
        if dets_p[0][b] is None:
            continue
        for i, (c, q) in enumerate(zip(dets_p[0][b].numpy(), dets_p[2][b].numpy())):
            cell = (int(c), int(q[0] // 4), int(q[1] // 4))
            if (b,) + cell not in only:
                continue
            ref_n += 1
            j = cells.get(cell)
            if j is None:
                missed += 1
                continue
            vd.append(float(np.abs(vv[j] - dets_p[3][b][i].numpy()).max()))
            if raws[b]['kept'][i] and fs[sl][j] < 0.1:
                d = np.abs(box_params(xs[sl][j:j + 1]) - box_params(raws[b]['x'][i:i + 1]))[0]
                d[0] = min(d[0], 2 * np.pi - d[0])
                boxd.append(d.max())
    vd, boxd = np.array(vd), np.array(boxd)
    return {'planted': ref_n, 'missed': missed, 'vert_p50': float(np.median(vd)), 'vert_max': float(vd.max()), 'boxes': len(boxd),
            'box_p50': float(np.median(boxd)), 'box_p90': float(np.percentile(boxd, 90)), 'box_max': float(boxd.max()),
            'within_1e-4': float((boxd <= 1e-4).mean())}


def stage(name):
    return 'backbone' if name.startswith('backbone') or name == 'input' else 'heads' if name.startswith('heads') else 'neck'


REG = (1, 2)            # head branches offset_fr_main, main_offset (the two Model.inference reads at the peaks; vertex_offset is never read)
VARIANTS = [
    ('all stages fp16 (emulation of the product)', lambda n, p, np_: True),
    ('fp32 backbone', lambda n, p, np_: stage(n) != 'backbone'),
    ('fp32 neck', lambda n, p, np_: stage(n) != 'neck'),
    ('fp32 heads', lambda n, p, np_: stage(n) != 'heads'),
    ('fp32 backbone + neck', lambda n, p, np_: stage(n) == 'heads'),
    ('fp32 the two regression branches only', lambda n, p, np_: not (stage(n) == 'heads' and np_ > 1 and p in REG)),
    ('fp32 everything but the two regression branches', lambda n, p, np_: stage(n) == 'heads' and np_ > 1 and p in REG),
    ('fp32 neck + the two regression branches', lambda n, p, np_: not (stage(n) == 'neck' or (stage(n) == 'heads' and np_ > 1 and p in REG))),
    ('fp32 everything (the verification mode)', None),
]

xd = x.to(dev)
rows = []
prod = [l.clone() for l in m.forward_logits(xd)]
rows.append(('PRODUCT path (fp16 storage, MFMA kernels)', stats(prod)))
m.forward_logits_fp32(xd)                 # builds the executor
ex = m._verify[1]
for name, f in VARIANTS:
    lgv = ex.forward(xd, m._head_channels, fp16=f)
    rows.append((name, stats(lgv)))
    print(name, rows[-1][1], flush=True)
m.release_verify()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'r04_error_apportioning.txt'), 'w') as fh:
    fh.write(__doc__.strip() + '\n\nDLA-34, %d benchmark images x %d planted cuboids, bench.py synthetic weights (seed 1, heat bias -6).\n\n' % (k, planted))
    fh.write('%-52s %8s %8s %9s %9s %9s %9s %8s %6s\n' % ('network precision', 'vert p50', 'vert max', 'box p50', 'box p90', 'box max', '<=1e-4', 'boxes', 'missed'))
    for name, s in rows:
        fh.write('%-52s %8.4f %8.4f %9.2e %9.2e %9.2e %8.1f%% %8d %6d\n' % (name, s['vert_p50'], s['vert_max'], s['box_p50'], s['box_p90'], s['box_max'],
                                                                          100 * s['within_1e-4'], s['boxes'], s['missed']))
print(open(os.path.join(ROOT, 'gpurun_out', 'r04_error_apportioning.txt')).read())
