"""Graph re-launch stress (round 3): one context, graphs of different structure (fp32 input: with the NHWC4 conversion node;
preloaded uint8 input: without it) and different input addresses, captured and re-launched in the orders that failed before
(n, n, n-1 -> relaunch first; n, n-1, n -> relaunch second) and in a seeded random order; every result is compared with the
eager replay bit for bit.  Progress in gpurun_out/graph_stress.log."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtm3d_amd
from rtm3d_amd import weights, preprocess


def run(say, bb='RESNET-18', B=3, H=128, W=256, rounds=40):
    dev = torch.device('cuda', 0)
    cfg = rtm3d_amd.kitti_config(bb)
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5)
    mg = rtm3d_amd.create_model(cfg).to(dev).eval(); mg.load_state_dict(sd); mg.use_graph = True
    me = rtm3d_amd.create_model(cfg).to(dev).eval(); me.load_state_dict(sd); me.use_graph = False
    xs = [weights.synth_images(B, H, W, seed=20 + i).to(dev) for i in range(3)]
    refs = [[t.clone() for t in me.forward_logits(x)] for x in xs]
    rng = np.random.Generator(np.random.PCG64(10))
    sizes = ((H * 3 // 4, W - 6), (H, W), (H // 2, W // 3))
    sets = [[torch.from_numpy(rng.integers(0, 256, size=(sizes[j % 3][0], sizes[j % 3][1], 3), dtype=np.uint8)).to(dev) for j in range(B)]
            for _ in range(2)]
    refp = []
    for imgs in sets:
        preprocess.preprocess_batch(imgs, (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, model=me)
        refp.append([t.clone() for t in me.forward_logits(None, preloaded=(B, H, W))])
    torch.cuda.synchronize()
    bad = 0

    def step(kind, i, outs):
        nonlocal bad
        say('begin %s %d outs=%s' % (kind, i, 'reuse' if isinstance(outs, str) else 'mine'))
        if kind == 'x':
            lg, ref = mg.forward_logits(xs[i], out=outs), refs[i]
        else:
            preprocess.preprocess_batch(sets[i], (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, model=mg)
            lg, ref = mg.forward_logits(None, preloaded=(B, H, W), out=outs), refp[i]
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(lg, ref))
        bad += 0 if ok else 1
        say('%s %s' % ('same     ' if ok else 'DIFFERENT', mg._plan_for(B, H, W, dev).graph_stats()))

    for kind, i in (('x', 0), ('x', 1), ('p', 0), ('p', 0), ('x', 0), ('p', 1), ('x', 1)):      # n, n, n-1 -> relaunch the first
        step(kind, i, 'reuse')
    mine = [torch.empty_like(t) for t in refs[0]]
    for kind, i in (('x', 0), ('p', 0), ('x', 2), ('p', 1), ('x', 0)):                          # n, n-1, n -> relaunch the second
        step(kind, i, mine)
    r = np.random.Generator(np.random.PCG64(3))
    for _ in range(rounds):
        kind = 'x' if r.random() < 0.5 else 'p'
        step(kind, int(r.integers(0, 3 if kind == 'x' else 2)), 'reuse' if r.random() < 0.5 else mine)
    return bad


if __name__ == '__main__':
    os.makedirs('gpurun_out', exist_ok=True)
    LOG = open('gpurun_out/graph_stress.log', 'w')

    def say(s):
        LOG.write(s + '\n'); LOG.flush(); os.fsync(LOG.fileno())
    bad = run(say)
    say('done RESNET-18: %d mismatches' % bad)
    bad2 = run(say, 'DLA-34', 1, 64, 128)
    say('done DLA-34: %d mismatches' % bad2)
    sys.exit(1 if bad or bad2 else 0)
