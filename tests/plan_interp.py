"""CPU interpreter of the plan IR (rtm3d_amd/plan.py) - TEST INFRASTRUCTURE, not a product path.

Executes the recorded ops with PyTorch-CPU on padded NHWC buffers exactly as the HIP runtime
addresses them (channel slices, tap offsets, sub-pixel phases, zero borders), so the host-side
graph wiring and BatchNorm folding can be validated against the oracle without a GPU.
With ``half=True`` weights and stored activations are rounded to fp16 (fp32 accumulate), which
predicts the numerical error of the fp16 MFMA path.
"""
import numpy as np
import torch


def run_plan(plan, x_nchw, half=False, prefill=None, yx=None):
    """prefill: {tensor id: (B, H, W, C) array} written into the interior of plan tensors before the first op (patch plans);
    yx: (B, 2) peak (y, x) per slot for `patch_mask` ops, -1 = empty slot."""
    B = plan.B
    rnd = (lambda t: t.half().float()) if half else (lambda t: t)
    bufs = []
    for t in plan.tensors:
        bufs.append(torch.zeros(B, t['H'] + 2 * t['pad'], t['W'] + 2 * t['pad'], t['C']))
    for tid, arr in (prefill or {}).items():
        t = plan.tensors[tid]
        bufs[tid][:, t['pad']:t['pad'] + t['H'], t['pad']:t['pad'] + t['W']] = rnd(torch.as_tensor(arr, dtype=torch.float32))
    outs = [None] * 4

    def view(s, extra=0):
        t = plan.tensors[s.tid]
        return bufs[s.tid], t['pad'], t['H'], t['W']

    for op in plan.ops:
        if op['op'] == 'input4':
            o = op['out']
            buf, P, H, W = view(o)
            buf[:, P:P + H, P:P + W, 0:3] = rnd(x_nchw.permute(0, 2, 3, 1))
        elif op['op'] == 'conv':
            Hm, Wm, s, sc = op['Hm'], op['Wm'], op['in_stride'], op['out_scale']
            for g in range(op['groups']):
                inp = op['inp'][g]
                ibuf, Pi, Hi, Wi = view(inp)
                acc = torch.zeros(B, Hm, Wm, op['cout'])
                wg = rnd(torch.from_numpy(op['w'][g]))                            # (taps, cout, cin)
                for t, (dy, dx) in enumerate(op['taps'][g]):
                    assert -Pi <= dy and (Hm - 1) * s + dy < Hi + Pi and -Pi <= dx and (Wm - 1) * s + dx < Wi + Pi, op['name']
                    # (rtm3d_conv_desc.tap_dc: a tap may name another channel slice of the input tensor; cin channels each)
                    c0 = inp.coff + (op['tap_dc'][g][t] if 'tap_dc' in op else 0)
                    cw = op['cin'] if 'tap_dc' in op else inp.C
                    assert c0 >= 0 and c0 + cw <= ibuf.shape[3], op['name']
                    xs = ibuf[:, Pi + dy: Pi + dy + (Hm - 1) * s + 1: s, Pi + dx: Pi + dx + (Wm - 1) * s + 1: s, c0:c0 + cw]
                    acc += xs @ wg[t].T
                acc = acc + torch.from_numpy(op['bias'][g])
                oy, ox = op['out_off'][g]
                r = op['res'][g]
                if r is not None:
                    rbuf, Pr, Hr, Wr = view(r)
                    acc = acc + rbuf[:, Pr + oy: Pr + oy + (Hm - 1) * sc + 1: sc, Pr + ox: Pr + ox + (Wm - 1) * sc + 1: sc, r.coff:r.coff + r.C]
                if op['relu']:
                    acc = acc.relu()
                if op['out_nchw']:
                    outs[op['out_nchw'] - 1] = acc.permute(0, 3, 1, 2).contiguous()
                else:
                    o = op['out'][g]
                    obuf, Po, Ho, Wo = view(o)
                    obuf[:, Po + oy: Po + oy + (Hm - 1) * sc + 1: sc, Po + ox: Po + ox + (Wm - 1) * sc + 1: sc, o.coff:o.coff + o.C] = rnd(acc)
        elif op['op'] == 's2d_copy':
            # the second, space-to-depth output of a feature's producer (RealizedPlan._neck_up_folds): pixel (y, x) of `src` ->
            # pixel (y >> 1, x >> 1), channels coff + ((y & 1) * 2 + (x & 1)) * C + c of tensor `tid`
            src = op['src']
            sbuf, Ps, Hs, Ws = view(src)
            dbuf = bufs[op['tid']]
            Pd = plan.tensors[op['tid']]['pad']
            v = sbuf[:, Ps:Ps + Hs, Ps:Ps + Ws, src.coff:src.coff + src.C]
            for py in range(2):
                for px in range(2):
                    c0 = op['coff'] + (py * 2 + px) * src.C
                    dbuf[:, Pd:Pd + Hs // 2, Pd:Pd + Ws // 2, c0:c0 + src.C] = v[:, py::2, px::2]
        elif op['op'] == 'zero_slice':
            # (test only: the ordinary copy of a feature that exists only as its space-to-depth copy on the device)
            sl = op['slice']
            bufs[sl.tid][..., sl.coff:sl.coff + sl.C] = 0
        elif op['op'] == 'maxpool_s2d':
            # rtm3d_op_maxpool_s2d: max over the four phase slices of one pixel
            ibuf, Pi = bufs[op['tid']], plan.tensors[op['tid']]['pad']
            o = op['out']
            obuf, Po, Ho, Wo = view(o)
            v = torch.stack([ibuf[:, Pi:Pi + Ho, Pi:Pi + Wo, op['coff'] + ph * o.C: op['coff'] + (ph + 1) * o.C] for ph in range(4)], 0).max(0).values
            obuf[:, Po:Po + Ho, Po:Po + Wo, o.coff:o.coff + o.C] = v
        elif op['op'] == 'headout':
            i = op['inp']
            ibuf, Pi, Hi, Wi = view(i)
            for h in range(len(op['w'])):
                xin = ibuf[:, Pi - 1:Pi + Hi + 1, Pi - 1:Pi + Wi + 1, h * 256:(h + 1) * 256].permute(0, 3, 1, 2)
                outs[h] = torch.nn.functional.conv2d(xin, rnd(torch.from_numpy(op['w'][h])), torch.from_numpy(op['bias'][h]))
        elif op['op'] == 'maxpool':
            i, o = op['inp'], op['out']
            ibuf, Pi, Hi, Wi = view(i)
            obuf, Po, Ho, Wo = view(o)
            k, s, p = op['k'], op['stride'], op['pad']
            assert p <= Pi
            win = ibuf[:, Pi - p: Pi - p + (Ho - 1) * s + k, Pi - p: Pi - p + (Wo - 1) * s + k, i.coff:i.coff + i.C]
            y = torch.nn.functional.max_pool2d(win.permute(0, 3, 1, 2), k, s)
            obuf[:, Po:Po + Ho, Po:Po + Wo, o.coff:o.coff + o.C] = y.permute(0, 2, 3, 1)
        elif op['op'] == 'patch_mask':
            buf, Pp, S, _ = view(op['t'])
            Hm, Wm = plan.map_hw
            for s_ in range(B):
                py, px = int(yx[s_][0]), int(yx[s_][1])
                if py < 0:
                    continue
                for i in range(S):
                    for j in range(S):
                        y, x = py + i - op['origin'], px + j - op['origin']
                        if not (0 <= y < Hm and 0 <= x < Wm):
                            buf[s_, i, j] = 0
        elif op['op'] == 'softmax':
            zi, zo = op['z_in'], op['z_out']
            zb, Pz, H, W = view(zi)
            acc = zb[:, Pz:Pz + H, Pz:Pz + W, zi.coff:zi.coff + zi.C].clone()
            for u in op['us']:
                ub, Pu, Hu, Wu = view(u)
                uu = ub[:, Pu:Pu + H, Pu:Pu + W, u.coff:u.coff + u.C]
                sm = torch.softmax(uu.reshape(B, H * W, -1), dim=1).reshape(B, H, W, -1)
                acc = acc + uu * sm
            ob, Po, _, _ = view(zo)
            ob[:, Po:Po + H, Po:Po + W, zo.coff:zo.coff + zo.C] = rnd(acc)
        else:
            raise AssertionError(op['op'])
    # borders must still be zero (the kernels rely on it)
    for t, b in zip(plan.tensors, bufs):
        P = t['pad']
        if P:
            assert float(b[:, :P].abs().max()) == 0 and float(b[:, :, :P].abs().max()) == 0
            assert float(b[:, -P:].abs().max()) == 0 and float(b[:, :, -P:].abs().max()) == 0

    def fetch(s):
        t = plan.tensors[s.tid]
        P = t['pad']
        return bufs[s.tid][:, P:P + t['H'], P:P + t['W'], s.coff:s.coff + s.C].permute(0, 3, 1, 2).contiguous()
    return outs, fetch
