"""fp32 verification executor of a plan (SURVEY.md H2 regime ii; include/rtm3d_hip.h "fp32 verification executor").

The product path (``plan.RealizedPlan`` -> ``rtm3d_forward``) stores activations and weights in fp16.  ``VerifyPlanF32``
replays the SAME ``plan.Plan`` - the op list ``build_plan`` records before any kernel is chosen: tap tables, channel
slices instead of ``torch.cat``, sub-pixel phases of the transposed convolutions, folded BatchNorm, composed 1x1 pairs -
on padded NHWC fp32 tensors through ``rtm3d_verify_{conv,maxpool,softmax_fuse}_f32`` (fp32 weights, fp64 accumulation).
Its logits fed to the product's own ``rtm3d_decode2d`` / ``rtm3d_decode3d_slots`` give the end-to-end answer to
"do the device's boxes match the reference's fp32 CPU path" (models/model.py:20-27 -> :29-75 ->
utils/model_utils.py:264-312) without the fp16 storage error in the way.

Verification only: about a hundred times slower than the MFMA path, used by ``Model.forward_logits_fp32``, the parity
tests and ``bench.py``'s ``parity`` block; ``Model.forward`` never takes it.  PyTorch only owns the device buffers.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class VerifyPlanF32(object):
    def __init__(self, plan, device):
        self.lib = _lib.load()
        self.plan = plan
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('VerifyPlanF32 runs on an AMD GPU only; the CPU interpreter of the plan is tests/plan_interp.py')
        B = plan.B
        with torch.cuda.device(self.device):
            # zero borders are written once here and never touched by the kernels (they implement the zero padding)
            self.bufs = [torch.zeros(B, t['H'] + 2 * t['pad'], t['W'] + 2 * t['pad'], t['C'], dtype=torch.float32, device=self.device)
                         for t in plan.tensors]
            self._w = {}
            for k, op in enumerate(plan.ops):
                if op['op'] == 'conv':
                    # (groups, taps, cout, cin) -> per group [taps][cin][cout]
                    self._w[k] = [(self._dev(np.ascontiguousarray(op['w'][g].transpose(0, 2, 1))), self._dev(op['bias'][g]))
                                  for g in range(op['groups'])]
                elif op['op'] == 'headout':
                    ws = []
                    for w, b in zip(op['w'], op['bias']):
                        co, ci = w.shape[0], w.shape[1]
                        ws.append((self._dev(np.ascontiguousarray(w.reshape(co, ci, 9).transpose(2, 1, 0))), self._dev(b)))
                    self._w[k] = ws
            self._ws = None

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)

    def _vt(self, s):
        t = self.plan.tensors[s.tid]
        v = _lib.VTensor()
        v.d = self.bufs[s.tid].data_ptr()
        v.Hp, v.Wp, v.C, v.P, v.coff = t['H'] + 2 * t['pad'], t['W'] + 2 * t['pad'], t['C'], t['pad'], s.coff
        return v

    def _fp16_parts(self, op, fp16):
        """[(lo, hi)] output-channel ranges (relative to each output slice) of `op` that the emulation stores as fp16.
        `fp16(name, part, nparts)`: the plan's four-branch head ops are asked per branch (the fused first head conv's 256-channel
        chunks, the groups of the second, the heads of the logit convs), every other op once."""
        if fp16 is None:
            return []
        if op['op'] == 'headout':
            return [(h, h + 1) for h in range(len(op['w'])) if fp16(op['name'], h, len(op['w']))]       # (head indices)
        if op['op'] == 'conv' and op['groups'] == 1 and op['cout'] % 256 == 0 and op['cout'] > 256 and op['name'].startswith('heads'):
            n = op['cout'] // 256
            return [(g * 256, g * 256 + 256) for g in range(n) if fp16(op['name'], g, n)]
        if op['op'] == 'conv' and op['groups'] > 1 and op['name'].startswith('heads'):
            return [(g, g + 1) for g in range(op['groups']) if fp16(op['name'], g, op['groups'])]              # (group indices)
        return [(0, None)] if fp16(op['name'], 0, 1) else []

    def forward(self, x, head_channels, fp16=None):
        """x: (B, 3, H, W) fp32 CUDA tensor -> tuple of fp32 NCHW logit maps (models/model.py:21-23).
        fp16 (error apportioning, tools/gpu_error_apportioning.py): callable (op name, part, parts) -> bool; the ops / head
        branches it selects are EMULATED in the product's storage precision - weights rounded to fp16, the written activations
        rounded to fp16 after the op (accumulation stays fp64 -> fp32): a mixed-precision replay that tells which stage's
        fp16 storage carries the end-to-end error.  None: everything fp32 (the verification mode)."""
        P, lib = self.plan, self.lib
        B = P.B
        if tuple(x.shape) != (B, 3, P.H, P.W) or not x.is_cuda or x.dtype != torch.float32:
            raise ValueError('VerifyPlanF32.forward: expected a (%d, 3, %d, %d) fp32 CUDA tensor' % (B, P.H, P.W))
        dev = self.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            outs = [torch.empty(B, c, P.H // 4, P.W // 4, dtype=torch.float32, device=dev) for c in head_channels]
            for k, op in enumerate(P.ops):
                kind = op['op']
                if kind == 'input4':
                    o = op['out']
                    t = P.tensors[o.tid]
                    pad = t['pad']
                    # layout change only (NCHW -> the NHWC4 operand of the stem, 4th channel stays zero)
                    self.bufs[o.tid][:, pad:pad + t['H'], pad:pad + t['W'], 0:3] = x.permute(0, 2, 3, 1)
                    if fp16 is not None and fp16(op['name'], 0, 1):
                        self.bufs[o.tid].copy_(self.bufs[o.tid].half().float())
                elif kind == 'conv':
                    parts = self._fp16_parts(op, fp16)
                    for g in range(op['groups']):
                        d = _lib.VConvDesc()
                        d.inp = self._vt(op['inp'][g])
                        if op['out_nchw']:
                            d.out.d = outs[op['out_nchw'] - 1].data_ptr()
                            d.out_nchw_f32, d.out_H, d.out_W = 1, op['out_hw'][0], op['out_hw'][1]
                        else:
                            d.out = self._vt(op['out'][g])
                        if op['res'][g] is not None:
                            d.res = self._vt(op['res'][g])
                        w, b = self._w[k][g]
                        if parts:
                            # fp16 weights for the emulated part: whole array (lo, None), 256-column chunks of a fused conv, or this group
                            if op['groups'] > 1 and op['name'].startswith('heads'):
                                if (g, g + 1) in parts:
                                    w = w.half().float()
                            else:
                                w = w.clone()
                                for lo, hi in parts:
                                    w[..., lo:hi] = w[..., lo:hi].half().float()
                        d.d_w, d.d_bias = w.data_ptr(), b.data_ptr()
                        d.B, d.Hm, d.Wm, d.in_stride, d.out_scale = B, op['Hm'], op['Wm'], op['in_stride'], op['out_scale']
                        d.out_oy, d.out_ox = op['out_off'][g]
                        d.cin, d.cout, d.ntaps, d.relu = op['cin'], op['cout'], len(op['taps'][g]), int(bool(op['relu']))
                        for t, (dy, dx) in enumerate(op['taps'][g]):
                            d.tap_dy[t], d.tap_dx[t] = dy, dx
                        _lib.check(lib.rtm3d_verify_conv_f32(stream, ctypes.byref(d)), 'verify_conv_f32(%s)' % op['name'])
                        if parts:
                            torch.cuda.current_stream(dev).synchronize()          # `w` may be a temporary
                    if parts and not op['out_nchw']:
                        if op['groups'] > 1 and op['name'].startswith('heads'):
                            for g, _ in parts:
                                o = op['out'][g]
                                v = self.bufs[o.tid][..., o.coff:o.coff + o.C]
                                v.copy_(v.half().float())
                        else:
                            o = op['out'][0]
                            for lo, hi in parts:
                                a, bnd = o.coff + lo, o.coff + (o.C if hi is None else hi)
                                v = self.bufs[o.tid][..., a:bnd]
                                v.copy_(v.half().float())
                elif kind == 'headout':
                    hparts = [h for h, _ in self._fp16_parts(op, fp16)]
                    taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
                    for h, (w, b) in enumerate(self._w[k]):
                        d = _lib.VConvDesc()
                        d.inp = self._vt(op['inp'])
                        d.inp.coff = op['inp'].coff + h * 256
                        d.out.d = outs[h].data_ptr()
                        d.out_nchw_f32, d.out_H, d.out_W = 1, P.H // 4, P.W // 4
                        if h in hparts:
                            w = w.half().float()
                        d.d_w, d.d_bias = w.data_ptr(), b.data_ptr()
                        d.B, d.Hm, d.Wm, d.in_stride, d.out_scale = B, P.H // 4, P.W // 4, 1, 1
                        d.cin, d.cout, d.ntaps, d.relu = 256, int(w.shape[2]), 9, 0
                        for t, (dy, dx) in enumerate(taps):
                            d.tap_dy[t], d.tap_dx[t] = dy, dx
                        _lib.check(lib.rtm3d_verify_conv_f32(stream, ctypes.byref(d)), 'verify_conv_f32(%s)' % op['name'])
                        if hparts:
                            torch.cuda.current_stream(dev).synchronize()
                elif kind == 'maxpool':
                    i, o = self._vt(op['inp']), self._vt(op['out'])
                    Ho, Wo = P.dims(op['out'])
                    _lib.check(lib.rtm3d_verify_maxpool_f32(stream, ctypes.byref(i), ctypes.byref(o), B, Ho, Wo, op['inp'].C,
                                                            op['k'], op['stride'], op['pad']), 'verify_maxpool_f32')
                elif kind == 'softmax':
                    zi, zo = self._vt(op['z_in']), self._vt(op['z_out'])
                    n_u = len(op['us'])
                    us = (_lib.VTensor * 3)()
                    for j, u in enumerate(op['us']):
                        us[j] = self._vt(u)
                    H, W = P.dims(op['z_in'])
                    C = op['z_in'].C
                    need = int(lib.rtm3d_verify_softmax_workspace_bytes(B, C, n_u))
                    if self._ws is None or self._ws.numel() < need:
                        self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
                    _lib.check(lib.rtm3d_verify_softmax_fuse_f32(stream, ctypes.byref(zi), ctypes.byref(zo), n_u, us, B, H, W, C,
                                                                 self._ws.data_ptr()), 'verify_softmax_fuse_f32')
                    if self._fp16_parts(op, fp16):
                        o = op['z_out']
                        v = self.bufs[o.tid][..., o.coff:o.coff + o.C]
                        v.copy_(v.half().float())
                else:
                    raise AssertionError('VerifyPlanF32: unknown op %r' % kind)
        return tuple(outs)

    def range_report(self, realized=None):
        """After ``forward``: the largest |value| of every tensor slice an op of the plan writes (fp32 run) and of every folded
        weight array, against the fp16 range the product path stores them in.  realized: the RealizedPlan of the same shape -
        its level rewrites store weights this executor never sees (the neck fold's composed taps A_up W and bias A_bias + b, the
        project folds' summed biases): their rows are added as 'weight (realized)' / 'bias (realized)', and activation rows whose
        tensor the realized plan no longer writes (the neck's `up` maps, maps that exist only as their space-to-depth copy) are
        tagged 'materialised': False (their values still bound what the fused kernels accumulate in fp32).  The reference runs in fp32
        (models/model.py:20-27) and cannot overflow; the product stores activations and weights as fp16 (max 65504) with no
        clamp: a checkpoint whose activations exceed that yields inf / NaN logits there (which rtm3d_decode2d handles like the
        reference's own NaN / Inf: tests/test_gpu_parity.py).  Rows: {'what': 'activation' | 'weight', 'op', 'tensor', 'max_abs',
        'headroom' (65504 / max_abs), 'overflow', 'order' (position in the plan)}, largest first."""
        FP16_MAX = 65504.0
        P = self.plan
        names = {(s.tid, s.coff, s.C): n for n, s in P.named.items()}
        rows = []

        def act(op_name, s):
            if s is None:
                return
            t = P.tensors[s.tid]
            pad = t['pad']
            v = float(self.bufs[s.tid][:, pad:pad + t['H'], pad:pad + t['W'], s.coff:s.coff + s.C].abs().max())
            rows.append({'what': 'activation', 'order': len(rows), 'op': op_name, 'slice': (s.tid, s.coff, s.C),
                         'tensor': names.get((s.tid, s.coff, s.C), 'tensor%d[%d:%d]' % (s.tid, s.coff, s.coff + s.C)),
                         'max_abs': v, 'headroom': FP16_MAX / v if v > 0 else float('inf'), 'overflow': not (v <= FP16_MAX)})

        for op in P.ops:
            kind = op['op']
            if kind == 'conv':
                for o in op['out']:
                    act(op['name'], o)
                w = float(np.abs(op['w']).max())
                rows.append({'what': 'weight', 'order': len(rows), 'op': op['name'], 'tensor': 'folded weights', 'max_abs': w,
                             'headroom': FP16_MAX / w if w > 0 else float('inf'), 'overflow': not (w <= FP16_MAX)})
            elif kind == 'headout':
                w = max(float(np.abs(x).max()) for x in op['w'])
                rows.append({'what': 'weight', 'order': len(rows), 'op': op['name'], 'tensor': 'folded weights', 'max_abs': w,
                             'headroom': FP16_MAX / w if w > 0 else float('inf'), 'overflow': not (w <= FP16_MAX)})
            elif kind == 'maxpool':
                act(op['name'], op['out'])
            elif kind == 'softmax':
                act(op['name'], op['z_out'])
        if realized is not None:
            for r in rows:
                if r['what'] == 'activation':
                    r['materialised'] = r['slice'] not in realized.unwritten
            for wr in realized.weight_ranges:
                for what, key in (('weight (realized)', 'max_abs_w'), ('bias (realized)', 'max_abs_bias')):
                    v = wr[key]
                    rows.append({'what': what, 'order': len(rows), 'op': wr['op'], 'tensor': 'as stored by the recorded op', 'max_abs': v,
                                 'headroom': FP16_MAX / v if v > 0 else float('inf'),
                                 'overflow': what.startswith('weight') and not (v <= FP16_MAX)})      # biases stay fp32
        rows.sort(key=lambda r: -r['max_abs'] if r['max_abs'] == r['max_abs'] else -float('inf'))
        return rows

    def fetch(self, name):
        """Stage output by its plan name (e.g. 'z', 'feat3') as fp32 NCHW, for parity tests."""
        s = self.plan.named[name]
        t = self.plan.tensors[s.tid]
        pad = t['pad']
        return self.bufs[s.tid][:, pad:pad + t['H'], pad:pad + t['W'], s.coff:s.coff + s.C].permute(0, 3, 1, 2).contiguous()
