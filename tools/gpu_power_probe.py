"""Is the bs=32 forward limited by average power?  Device time of one forward (hipEvents) when forwards run back to back against
forwards separated by idle gaps of 5 / 20 / 50 ms, and of the head convs alone in both regimes (round 5)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtm3d_amd
from rtm3d_amd import weights
dev = torch.device('cuda', 0)
bb = 'DLA-34'
m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to(dev).eval()
m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0))
B, H, W = 32, 384, 1280
x = weights.synth_images(B, H, W, seed=1234).to(dev)
for _ in range(20):
    m.forward_logits(x, out='reuse')
torch.cuda.synchronize()
for gap_ms in (0, 5, 20, 50, 0):
    ts = []
    for _ in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.forward_logits(x, out='reuse'); e1.record()
        if gap_ms:
            torch.cuda.synchronize(); time.sleep(gap_ms * 1e-3)
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts[5:])
    print('idle gap %2d ms between forwards: forward %.3f ms (median of 20; min %.3f max %.3f)' % (gap_ms, v[len(v) // 2], v[0], v[-1]))
