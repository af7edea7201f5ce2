"""Micro-benchmark + cross-check of the conv kernel variants on one layer shape (GPU only)."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtm3d_amd import plan as plan_mod

def one(B, H, W, cin, cout, k, dil, variant, groups=1, deconv=False, reps=5, seed=0, bn_tile=None, check=None):
    rng = np.random.default_rng(seed)
    P = plan_mod.Plan(B, H * 4, W * 4)
    pad = dil * (k - 1) // 2
    if deconv:
        x = P.tensor(H, W, cin, 1); y = P.tensor(2 * H, 2 * W, cout, 0)
        w = (rng.standard_normal((cin, cout, 4, 4)) * 0.05).astype(np.float32)
        P.deconv(x, y, w, name='deconv')
    elif groups == 1:
        x = P.tensor(H, W, cin, pad); y = P.tensor(H, W, cout, 1)
        w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
        P.conv(x, y, w, rng.standard_normal(cout).astype(np.float32), dil=dil, relu=True, name='conv')
    else:
        x = P.tensor(H, W, cin * groups, pad); y = P.tensor(H, W, cout * groups, 1)
        ws = [(rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32) for _ in range(groups)]
        bs = [rng.standard_normal(cout).astype(np.float32) for _ in range(groups)]
        P.grouped_conv([P.sub(x, g * cin, cin) for g in range(groups)], [P.sub(y, g * cout, cout) for g in range(groups)], ws, bs, dil=dil, relu=True, name='gconv')
    P.ops[-1]['variant'] = variant
    if bn_tile: P.ops[-1]['bn_tile'] = bn_tile
    R = plan_mod.RealizedPlan(P, 0)
    xin = rng.standard_normal((B, x.C, H, W)).astype(np.float32)
    from rtm3d_amd import _lib
    import ctypes
    _lib.check(R.lib.rtm3d_tensor_upload(R.ctx, R.tids[x.tid], 0, x.C, xin.ctypes.data_as(ctypes.c_void_p)))
    dummy = torch.zeros(16, device='cuda'); outs = [torch.zeros(16, device='cuda') for _ in range(4)]
    s = torch.cuda.current_stream().cuda_stream
    ms = []
    for _ in range(reps):
        info = R.forward_timed(s, dummy.data_ptr(), [o.data_ptr() for o in outs]); ms.append(info[0]['ms'])
    out = R.download(y)
    fl = info[0]['flops']
    best = min(ms)
    print('%-8s B=%d %dx%d cin=%d cout=%d k=%d d=%d g=%d variant=%s: best %.3f ms median %.3f  -> %.0f TFLOP/s' % (
        'deconv' if deconv else 'conv', B, H, W, cin, cout, k, dil, groups, variant, best, float(np.median(ms)), fl / best / 1e9))
    if check is not None:
        err = np.abs(out - check).max()
        print('   max |diff| vs other variant: %.4g (scale %.3g)' % (err, np.abs(check).max()))
    R.close()
    return out

if __name__ == '__main__':
    ap = argparse.ArgumentParser(); ap.add_argument('--quick', action='store_true'); a = ap.parse_args()
    B = 8 if a.quick else 32
    r0 = one(B, 96, 320, 256, 1024, 3, 6, 0)
    one(B, 96, 320, 256, 1024, 3, 6, 2, check=r0)
    r0 = one(B, 96, 320, 256, 256, 3, 1, 0, groups=4)
    one(B, 96, 320, 256, 256, 3, 1, 2, groups=4, check=r0)
    r0 = one(B, 48, 160, 256, 256, 4, 1, 0, deconv=True)
    one(B, 48, 160, 256, 256, 4, 1, 2, deconv=True, check=r0)
