"""Quick on-GPU diagnostic (not a test): per-stage errors of the HIP path vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtm3d_amd
from rtm3d_amd import weights, plan as plan_mod
from oracle import rtm3d_ref, decode3d_ref

def stage_check(bb, B=1, H=64, W=128, seed=3):
    sd = weights.synth_state_dict(bb, seed, 'trained', heat_bias=-3.0)
    x = weights.synth_images(B, H, W, seed=5)
    cfg = rtm3d_amd.kitti_config(bb)
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval(); m.load_state_dict(sd)
    t = time.time(); logits = m.forward_logits(x.cuda()); torch.cuda.synchronize(); print(bb, 'first forward %.2fs' % (time.time() - t))
    dets, lref, st = rtm3d_ref.model_forward(x, sd, bb, return_stages=True)
    plan = m._plan_for(B, H, W, torch.device('cuda', 0))
    # run the CPU interpreter of the same plan to localise a wrong op
    from tests.plan_interp import run_plan
    _, fetch = run_plan(plan.plan, x, half=True)
    names = list(plan.plan.named.keys())
    for nme in names:
        s = plan.plan.named[nme]
        g = plan.download(s); r = fetch(s).numpy()
        print('  %-10s max|gpu-interp|=%.4g  scale=%.3g' % (nme, np.abs(g - r).max(), np.abs(r).max()))
    for i in range(4):
        print('  logits[%d] max err vs oracle %.4g (scale %.3g)' % (i, (logits[i].cpu() - lref[i]).abs().max().item(), lref[i].abs().max().item()))
    return m, sd

for bb in ['DLA-34', 'RESNET-18']:
    stage_check(bb)

# decode2d on golden cases
from tests.golden.cases import DECODE2D_CASES, decode2d_inputs
g = np.load('tests/golden/decode2d_cases.npz')
cfg = rtm3d_amd.kitti_config('RESNET-18'); m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
for name in DECODE2D_CASES:
    th, tk, arrs = decode2d_inputs(name)
    m.config.DETECTOR.SCORE_THRESH, m.config.DETECTOR.TOPK_CANDIDATES = th, tk
    d = m.inference([torch.from_numpy(a).cuda() for a in arrs])
    n = g[name + '_det_n']
    ok = True
    for b in range(len(n)):
        if n[b] == 0:
            ok &= d[0][b] is None; continue
        if d[0][b] is None or len(d[0][b]) != n[b]:
            ok = False; print('   count mismatch', name, b, None if d[0][b] is None else len(d[0][b]), n[b]); continue
        for k, key in enumerate(['cls', 'score', 'mproj', 'verts', 'bbox']):
            a = d[k][b].cpu().numpy(); r = g['%s_det_%s_%d' % (name, key, b)]
            if not np.array_equal(a, r):
                ok = False; print('   mismatch', name, b, key, np.abs(a.astype(np.float64) - r).max(), (a != r).sum())
    print('decode2d', name, 'OK' if ok else 'FAIL')

# decode3d
g3 = np.load('tests/golden/decode3d_cases.npz')
t = time.time()
x, fun, nit, st = rtm3d_amd.model_utils.solve_boxes(g3['clses'], g3['uv'], g3['K'], g3['dim_ref'], g3['ref_loc'])
print('decode3d 64 objs %.3fs' % (time.time() - t), 'max|dx|', np.abs(x - g3['raw_x']).max(), 'kept equal', ((fun < 0.1) == (g3['raw_fun'] < 0.1)).all(),
      'max dx kept', np.abs(x - g3['raw_x'])[g3['raw_fun'] < 0.1].max())
t = time.time()
for _ in range(3):
    rtm3d_amd.model_utils.solve_boxes(np.tile(g3['clses'], 50), np.tile(g3['uv'], (50, 1, 1)), g3['K'], g3['dim_ref'], g3['ref_loc'])
print('decode3d 3200 objs: %.1f ms per call' % ((time.time() - t) / 3 * 1e3))
