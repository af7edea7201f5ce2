"""Mixed soak (round 3): ONE process alternates shapes, batch sizes and head modes at random for a given number of seconds - plans are
evicted and rebuilt (MAX_PLANS = 3), graphs captured and re-launched, the patch plan of the peaks-only heads comes and goes - and
every result must equal, bit for bit, what the same (shape, mode) produced the first time.
    python tools/gpu_mixed_soak.py [seconds] > gpurun_out/mixed_soak.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtm3d_amd
from rtm3d_amd import weights, distributed as rdist

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device('cuda', 0)
bb = 'DLA-34'
cfg = rtm3d_amd.kitti_config(bb)
m = rtm3d_amd.create_model(cfg).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-4.0))
shapes = [(1, 384, 1280), (8, 384, 1280), (2, 128, 256), (3, 64, 128), (1, 128, 256)]
xs = {s: weights.synth_images(s[0], s[1], s[2], seed=100 + i).to(dev) for i, s in enumerate(shapes)}
Ks = {s: torch.as_tensor(np.tile(weights.synth_intrinsics(), (s[0], 1)), dtype=torch.float64, device=dev) for s in shapes}
first = {}
rng = np.random.Generator(np.random.PCG64(5))
t0 = time.time()
it = bad = 0
counts = {}
while time.time() - t0 < budget:
    s = shapes[int(rng.integers(0, len(shapes)))]
    sparse = bool(rng.integers(0, 2))
    det, boxes, lg = m.detect3d(xs[s], Ks[s], sparse_heads=sparse)
    rec = rdist.pack_records(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, det.topk, boxes)
    torch.cuda.synchronize()
    key = (s, sparse)
    counts[key] = counts.get(key, 0) + 1
    if key not in first:
        first[key] = (rec.clone(), lg[0].clone())
    else:
        ok = torch.equal(rec, first[key][0]) and torch.equal(lg[0], first[key][1])
        if not ok:
            bad += 1
            print('MISMATCH at iteration %d: %s' % (it, key), flush=True)
    it += 1
    if it % 200 == 0:
        print('%6.0f s  %d iterations, %d mismatches, %d plans cached, %.1f GB allocated by torch' %
              (time.time() - t0, it, bad, len(m._plans), torch.cuda.memory_allocated(dev) / 1e9), flush=True)
free, total = torch.cuda.mem_get_info(dev)
print('done: %d iterations in %.0f s over %d (shape, mode) combinations, %d mismatches; device memory in use at the end %.1f GB' %
      (it, time.time() - t0, len(first), bad, (total - free) / 1e9))
for k in sorted(counts):
    print('  %-18s %-6s x %d  detections %d' % (k[0], 'sparse' if k[1] else 'dense', counts[k], int((first[k][0][..., 31] > 0).sum())))
sys.exit(1 if bad else 0)
