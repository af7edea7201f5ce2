import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import rtm3d_amd
from rtm3d_amd import weights
dev = torch.device('cuda', 0)
bb = 'DLA-34'
m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
B, H, W = 32, 384, 1280
x = weights.synth_images(B, H, W, seed=1234).to(dev)
plan = m._plan_for(B, H, W, dev)
for lanes in (1, 0, 1, 0):
    plan.set_lanes(lanes)
    for mode in ('loop', 'sync'):
        for _ in range(3):
            m.forward_logits(x, out='reuse')
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            m.forward_logits(x, out='reuse')
            if mode == 'sync':
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        print('lanes', lanes, mode, 'ms/forward %.3f' % ((time.perf_counter() - t) / 20 * 1e3))
