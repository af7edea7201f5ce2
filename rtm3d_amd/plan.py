"""Host-side plan builder: reference state-dict -> list of kernel launches for librtm3d_hip.so.

Orchestration only (BASELINE north_star: "host code stays Python for weight loading and
orchestration").  ``build_plan`` walks the architecture once and emits a small IR:

  tensors : padded NHWC fp16 activations  {H, W, C, pad}
  ops     : input4 | conv | headout | maxpool | softmax   with BatchNorm already folded into weights/bias

``realize`` packs the weights for the chosen kernel and records the plan into a runtime context
through the C ABI.  The IR is plain numpy so the wiring (channel slices instead of torch.cat,
sub-pixel phases of the transposed convolutions, BN folding) can be checked on a CPU-only
machine by tests/plan_interp.py.

Graph sources: models/nets/dla.py:186-210,322-332 (DLA-34 trees), models/nets/resnet.py:143-158,
200-211 (ResNet), models/nets/keypoint_fpn_fusion.py:35-69 (neck), models/nets/header.py:40-46.
"""
import ctypes
import math
import os

import numpy as np

from . import _lib
from .weights import parse_backbone, DLA34_CHANNELS, DLA34_LEVELS, RESNET_BLOCKS, head_table

V2_MIN_TILES = 200
FUSE_LEVEL_ENTRY = True   # DLA level2 entry: 2x2 max-pool + 1x1 project + 3x3 stride-2 conv in one launch (conv32s2_fused.hip)
FUSE_LEVEL_TAIL = True    # DLA level2 tail: tree2.conv2 + residual, the root 1x1 and the next level's 2x2 max-pool in one launch (conv64_root.hip)
FOLD_PROJECT = True       # DLA levels 3-5: a block's `project` 1x1 (on the pooled input) as extra K-steps of the block's second conv
FOLD_PROJECT_C128 = os.environ.get('RTM3D_FOLD_C128', '1') != '0'  # ... also where that conv would otherwise take conv128_halo (level3), which has no one-tap chunk: generic kernel
FOLD_NECK_UP = os.environ.get('RTM3D_FOLD_NECK_UP', '1') != '0'   # neck: proj+head 1x1 composed INTO the transposed conv in front of it (RealizedPlan._neck_up_folds)
FUSE_STEM = True        # DLA stem: base_layer + level0 in one launch (conv_stem_fused.hip); False = two launches (A/B, tests)
BN_EPS = 1e-4   # utils/torch_utils.py:79-81: initialize_weights sets eps=1e-4 on every BatchNorm2d


class Slice(object):
    """Channels [coff, coff+C) of a plan tensor."""
    def __init__(self, tid, coff, C):
        self.tid, self.coff, self.C = tid, coff, C


class Plan(object):
    def __init__(self, B, H, W):
        self.B, self.H, self.W = B, H, W
        self.tensors = []
        self.ops = []
        self.named = {}      # debug names -> Slice (stage outputs, for parity tests)

    # ---- IR construction
    def tensor(self, H, W, C, pad, name=None):
        self.tensors.append({'H': H, 'W': W, 'C': C, 'pad': pad})
        s = Slice(len(self.tensors) - 1, 0, C)
        if name:
            self.named[name] = s
        return s

    def sub(self, s, coff, C, name=None):
        r = Slice(s.tid, s.coff + coff, C)
        if name:
            self.named[name] = r
        return r

    def dims(self, s):
        t = self.tensors[s.tid]
        return t['H'], t['W']

    def conv(self, inp, out, w, bias, stride=1, dil=1, relu=False, res=None, name='', out_nchw=0, out_hw=None):
        """w: (cout, cin, k, k) fp32 with BN folded; padding = dil*(k-1)//2 (all reference convs)."""
        cout, cin, k, _ = w.shape
        assert inp.C == cin, (name, inp.C, cin)
        pad = dil * (k - 1) // 2
        taps = [(ky * dil - pad, kx * dil - pad) for ky in range(k) for kx in range(k)]
        wt = np.stack([w[:, :, ky, kx] for ky in range(k) for kx in range(k)], 0)[None]   # (1, taps, cout, cin)
        Hi, Wi = self.dims(inp)
        Hm, Wm = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        if out is not None:
            assert out.C == cout and self.dims(out) == (Hm, Wm), (name, out.C, cout, self.dims(out), (Hm, Wm))
            # a residual must not alias the output: the kernels pad a partial pixel tile with copies of its last pixel ("a
            # same-value write"), and a copy that loads the residual after another copy has stored would add it twice
            assert res is None or res.tid != out.tid or res.coff + res.C <= out.coff or out.coff + out.C <= res.coff, (name, 'residual aliases the output')
        self.ops.append({'op': 'conv', 'name': name, 'inp': [inp], 'out': [out], 'res': [res], 'Hm': Hm, 'Wm': Wm,
                         'in_stride': stride, 'out_scale': 1, 'cin': cin, 'cout': cout, 'groups': 1,
                         'taps': [taps], 'out_off': [(0, 0)], 'relu': relu, 'w': wt.astype(np.float32),
                         'bias': np.asarray(bias, np.float32)[None], 'out_nchw': out_nchw,
                         'out_hw': out_hw or (Hm, Wm)})

    def grouped_conv(self, inps, outs, ws, biases, dil=1, relu=False, name=''):
        """Same-shape 3x3 convs on different channel slices in one launch (the four head branches)."""
        G = len(inps)
        cout, cin, k, _ = ws[0].shape
        pad = dil * (k - 1) // 2
        taps = [(ky * dil - pad, kx * dil - pad) for ky in range(k) for kx in range(k)]
        wt = np.stack([np.stack([w[:, :, ky, kx] for ky in range(k) for kx in range(k)], 0) for w in ws], 0)
        Hm, Wm = self.dims(inps[0])
        self.ops.append({'op': 'conv', 'name': name, 'inp': list(inps), 'out': list(outs), 'res': [None] * G,
                         'Hm': Hm, 'Wm': Wm, 'in_stride': 1, 'out_scale': 1, 'cin': cin, 'cout': cout, 'groups': G,
                         'taps': [taps] * G, 'out_off': [(0, 0)] * G, 'relu': relu, 'w': wt.astype(np.float32),
                         'bias': np.stack([np.asarray(b, np.float32) for b in biases], 0), 'out_nchw': 0,
                         'out_hw': (Hm, Wm)})

    def deconv(self, inp, out, w, name=''):
        """ConvTranspose2d(k=4, s=2, p=1, bias=False) (models/nets/module.py:7-15), w: (cin, cout, 4, 4),
        as four sub-pixel phases: output row 2y+py receives input rows {y, y-1} (py=0, ky=1,3) or
        {y+1, y} (py=1, ky=0,2); same along x.  No zero insertion."""
        cin, cout = w.shape[0], w.shape[1]
        assert inp.C == cin and out.C == cout
        Hi, Wi = self.dims(inp)
        assert self.dims(out) == (2 * Hi, 2 * Wi)
        kd = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}     # phase -> [(k index, input offset)]
        taps, wts, offs = [], [], []
        for py in (0, 1):
            for px in (0, 1):
                tp, wp = [], []
                for ky, dy in kd[py]:
                    for kx, dx in kd[px]:
                        tp.append((dy, dx))
                        wp.append(w[:, :, ky, kx].T)           # (cout, cin)
                taps.append(tp); wts.append(np.stack(wp, 0)); offs.append((py, px))
        self.ops.append({'op': 'conv', 'name': name, 'inp': [inp] * 4, 'out': [out] * 4, 'res': [None] * 4,
                         'Hm': Hi, 'Wm': Wi, 'in_stride': 1, 'out_scale': 2, 'cin': cin, 'cout': cout, 'groups': 4,
                         'taps': taps, 'out_off': offs, 'relu': False, 'w': np.stack(wts, 0).astype(np.float32),
                         'bias': np.zeros((4, cout), np.float32), 'out_nchw': 0, 'out_hw': (2 * Hi, 2 * Wi)})

    def maxpool(self, inp, out, k, stride, pad, name=''):
        assert inp.C == out.C
        self.ops.append({'op': 'maxpool', 'name': name, 'inp': inp, 'out': out, 'k': k, 'stride': stride, 'pad': pad})

    def input_nhwc4(self, out, name='input'):
        """fp32 NCHW image -> 4-channel padded NHWC fp16 tensor (4th channel zero)."""
        assert out.C == 4 and out.coff == 0 and self.tensors[out.tid]['pad'] >= 4
        self.ops.append({'op': 'input4', 'name': name, 'out': out})

    def stem_mfma(self, out, w, bias, stride, name=''):
        """7x7 stem as a register-direct MFMA conv over the NHWC4 image (conv_smallc.hip)."""
        cout, cin, k, _ = w.shape
        assert cin == 3 and k == 7
        Ho, Wo = self.dims(out)
        x4 = self.tensor(Ho * stride, Wo * stride, 4, 4, name='input4')
        self.input_nhwc4(x4)
        w4 = np.zeros((cout, 4, k, k), np.float32)
        w4[:, :3] = w
        self.conv(x4, out, w4, bias, stride=stride, relu=True, name=name)

    def headout(self, inp, ws, biases, name='heads.out'):
        """The four final 3x3 convs (cout <= 16 each) in one halo-tile launch -> fp32 NCHW logits."""
        assert inp.coff == 0 and 1 <= len(ws) <= 4 and inp.C == 256 * len(ws)
        self.ops.append({'op': 'headout', 'name': name, 'inp': inp, 'w': [np.asarray(w, np.float32) for w in ws],
                         'bias': [np.asarray(b, np.float32) for b in biases]})

    def conv_taps(self, inps, outs, ws, biases, taps, Hm, Wm, relu=False, name='', out_nchw=0):
        """Convolution with an explicit tap list and iteration domain (no implied padding): output (y, x), y < Hm, x < Wm, reads
        input pixels (y + dy, x + dx) for (dy, dx) in taps.  inps / outs / ws / biases: one entry per group (same shapes);
        ws[g]: (cout, cin, len(taps)).  outs[g] None with out_nchw = k: fp32 NCHW result in output slot k - 1."""
        G = len(inps)
        cout, cin, nt = ws[0].shape
        assert nt == len(taps) and all(i.C == cin for i in inps)
        wt = np.stack([np.stack([w[:, :, t] for t in range(nt)], 0) for w in ws], 0)             # (G, taps, cout, cin)
        self.ops.append({'op': 'conv', 'name': name, 'inp': list(inps), 'out': list(outs), 'res': [None] * G, 'Hm': Hm, 'Wm': Wm,
                         'in_stride': 1, 'out_scale': 1, 'cin': cin, 'cout': cout, 'groups': G, 'taps': [list(taps)] * G,
                         'out_off': [(0, 0)] * G, 'relu': relu, 'w': wt.astype(np.float32),
                         'bias': np.stack([np.asarray(b, np.float32) for b in biases], 0), 'out_nchw': out_nchw, 'out_hw': (Hm, Wm)})

    def patch_mask(self, t, origin, name='patch_mask'):
        """Patch plans (build_peak_plan): zero the window positions of `t` that lie outside the map (they are the next conv's
        zero padding); position (i, j) of slot s is map pixel (y_s + i - origin, x_s + j - origin)."""
        self.ops.append({'op': 'patch_mask', 'name': name, 't': t, 'origin': origin})

    def softmax_fuse(self, z_in, z_out, us, name=''):
        self.ops.append({'op': 'softmax', 'name': name, 'z_in': z_in, 'z_out': z_out, 'us': list(us)})

    def total_flops(self):
        """Algorithmic FLOPs of the REFERENCE graph (2 x MAC of its conv layers): composed 1x1 pairs count as
        the two original layers, the NHWC4 stem as 3 input channels."""
        f = 0.0
        for op in self.ops:
            if op['op'] == 'conv' and 'ref_macs_per_px' in op:
                f += 2.0 * self.B * op['Hm'] * op['Wm'] * op['ref_macs_per_px']
            elif op['op'] == 'conv':
                cin = 3 if op['cin'] == 4 else op['cin']          # NHWC4 stem: the 4th channel is zero padding
                f += 2.0 * self.B * op['Hm'] * op['Wm'] * op['groups'] * cin * len(op['taps'][0]) * op['cout']
            elif op['op'] == 'headout':
                H, W = self.dims(op['inp'])
                f += 2.0 * self.B * H * W * 9 * 256 * sum(w.shape[0] for w in op['w'])
        return f


# ------------------------------------------------------------------------------------------
def _np(sd, key):
    return sd[key].detach().cpu().numpy().astype(np.float64)


_CACHE = None        # WeightCache of the build_plan / RealizedPlan in progress (single-threaded host code)


def fold_bn(sd, conv, bn=None):
    """conv weight (+bias) followed by eval-mode BatchNorm(eps=1e-4) -> (w', b') in fp32 (cached per state dict)."""
    if _CACHE is not None:
        return _CACHE.get('fold:%s|%s' % (conv, bn), lambda: _fold_bn(sd, conv, bn))
    return _fold_bn(sd, conv, bn)


def _fold_bn(sd, conv, bn=None):
    w = _np(sd, conv + '.weight')
    b = _np(sd, conv + '.bias') if (conv + '.bias') in sd else np.zeros(w.shape[0])
    if bn is not None:
        s = _np(sd, bn + '.weight') / np.sqrt(_np(sd, bn + '.running_var') + BN_EPS)
        w = w * s[:, None, None, None]
        b = (b - _np(sd, bn + '.running_mean')) * s + _np(sd, bn + '.bias')
    return w.astype(np.float32), b.astype(np.float32)


def _dla_block(P, sd, p, x, out, stride, residual, mid=None):
    """BasicBlock (models/nets/dla.py:86-100): relu(bn2(conv2(relu(bn1(conv1 x)))) + residual)."""
    Ho, Wo = P.dims(out)
    if mid is None:
        mid = P.tensor(Ho, Wo, out.C, 1)
    w, b = fold_bn(sd, p + '.conv1', p + '.norm1')
    P.conv(x, mid, w, b, stride=stride, relu=True, name=p + '.conv1')
    w, b = fold_bn(sd, p + '.conv2', p + '.norm2')
    P.conv(mid, out, w, b, relu=True, res=residual, name=p + '.conv2')


def _dla_tree1(P, sd, p, x, bottom, cat, cin, cout, stride, extra):
    """Level-1 Tree (models/nets/dla.py:186-206).  `cat` is the root's input: [x2 | x1 | children...];
    children were already written into their slices by the caller.  Returns nothing; root output is
    written to `extra['out']`."""
    if cin != cout:
        Hc, Wc = P.dims(cat)
        resid = P.tensor(Hc, Wc, cout, 0)
        w, b = fold_bn(sd, p + '.project.0', p + '.project.1')
        P.conv(bottom, resid, w, b, name=p + '.project')
    else:
        resid = bottom
    x2s, x1s = P.sub(cat, 0, cout), P.sub(cat, cout, cout)
    _dla_block(P, sd, p + '.tree1', x, x1s, stride, resid, mid=extra.get('mid'))
    _dla_block(P, sd, p + '.tree2', x1s, x2s, 1, x1s)
    w, b = fold_bn(sd, p + '.root.conv', p + '.root.norm')
    P.conv(cat, extra['out'], w, b, relu=True, name=p + '.root')


def _build_dla(P, sd, H, W, feat_out):
    ch = DLA34_CHANNELS
    t_base = P.tensor(H, W, ch[0], 1, name='base')
    w, b = fold_bn(sd, 'backbone.base_layer.0', 'backbone.base_layer.1')
    P.stem_mfma(t_base, w, b, 1, name='backbone.base_layer')
    t_l0 = P.tensor(H, W, ch[0], 1, name='level0')
    w, b = fold_bn(sd, 'backbone.level0.0', 'backbone.level0.1')
    P.conv(t_base, t_l0, w, b, relu=True, name='backbone.level0')
    t_l1 = P.tensor(H // 2, W // 2, ch[1], 1, name='level1')
    w, b = fold_bn(sd, 'backbone.level1.0', 'backbone.level1.1')
    P.conv(t_l0, t_l1, w, b, stride=2, relu=True, name='backbone.level1')
    x = t_l1
    for i in range(2, 6):
        p = 'backbone.level%d' % i
        cin, cout = ch[i - 1], ch[i]
        Hi, Wi = P.dims(x)
        Ho, Wo = Hi // 2, Wi // 2
        out = feat_out[i - 2]
        if DLA34_LEVELS[i] == 1:
            level_root = i > 2
            # (level_root: the first block's intermediate map lives in the tensor that holds the pooled input, so that the block's
            # `project` 1x1 can run as extra K-steps of its second conv: RealizedPlan._project_folds)
            cat = P.tensor(Ho, Wo, 2 * cout + ((cin + cout) if level_root else 0), 1)
            mid = None
            if level_root:
                bottom = P.sub(cat, 2 * cout, cin)
                mid = P.sub(cat, 2 * cout + cin, cout)
            else:
                bottom = P.tensor(Ho, Wo, cin, 0)
            P.maxpool(x, bottom, 2, 2, 0, name=p + '.downsample')
            _dla_tree1(P, sd, p, x, bottom, P.sub(cat, 0, 2 * cout + (cin if level_root else 0)), cin, cout, 2, {'out': out, 'mid': mid})
        else:
            # level-2 tree with level_root: final root input = [x2' | x1' | bottom | x1_outer]
            cat2t = P.tensor(Ho, Wo, 4 * cout + cin, 1)          # [x2' | x1' | bottom | x1_outer] + tree1.tree1's intermediate map
            cat2 = P.sub(cat2t, 0, 3 * cout + cin)
            bottom = P.sub(cat2, 2 * cout, cin)
            x1_outer = P.sub(cat2, 2 * cout + cin, cout)
            mid = P.sub(cat2t, 3 * cout + cin, cout)
            P.maxpool(x, bottom, 2, 2, 0, name=p + '.downsample')
            # (the outer tree's own `project` output is never consumed: models/nets/dla.py:195-202)
            cat1 = P.tensor(Ho, Wo, 2 * cout, 1)
            _dla_tree1(P, sd, p + '.tree1', x, bottom, cat1, cin, cout, 2, {'out': x1_outer, 'mid': mid})
            _dla_tree1(P, sd, p + '.tree2', x1_outer, x1_outer, cat2, cout, cout, 1, {'out': out})
        x = out


def _build_resnet(P, sd, H, W, depth, feat_out):
    t_c1 = P.tensor(H // 2, W // 2, 64, 1, name='conv1')
    w, b = fold_bn(sd, 'backbone.conv1', 'backbone.bn1')
    P.stem_mfma(t_c1, w, b, 2, name='backbone.conv1')
    x = P.tensor(H // 4, W // 4, 64, 1, name='pool')
    P.maxpool(t_c1, x, 3, 2, 1, name='backbone.maxpool')
    inpl = 64
    for li, (pl, nb) in enumerate(zip([64, 128, 256, 512], RESNET_BLOCKS[depth])):
        for bi in range(nb):
            p = 'backbone.layer%d.%d' % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            Hi, Wi = P.dims(x)
            Ho, Wo = Hi // stride, Wi // stride
            last = bi == nb - 1
            out = feat_out[li] if last else P.tensor(Ho, Wo, pl, 1)
            mid = P.tensor(Ho, Wo, pl, 1)
            w, b = fold_bn(sd, p + '.conv1', p + '.bn1')
            P.conv(x, mid, w, b, stride=stride, relu=True, name=p + '.conv1')
            if bi == 0 and (stride != 1 or inpl != pl):
                resid = P.tensor(Ho, Wo, pl, 0)
                w, b = fold_bn(sd, p + '.downsample.0', p + '.downsample.1')
                P.conv(x, resid, w, b, stride=stride, name=p + '.downsample')
            else:
                resid = x
            w, b = fold_bn(sd, p + '.conv2', p + '.bn2')
            P.conv(mid, out, w, b, relu=True, res=resid, name=p + '.conv2')
            x = out
            inpl = pl


def build_plan(state_dict, backbone, B, H, W, head_variant='rtm3d', cache=None, num_classes=3, dense_heads=None, header_num_conv=2):
    """state_dict: reference key names -> torch tensors.  H, W multiples of 32.  cache: a WeightCache of this state dict.
    dense_heads = k: only the first k head branches are evaluated on the whole map (1 = the heat map alone: the other
    branches then come from build_peak_plan at the detected peaks)."""
    global _CACHE
    _CACHE = cache
    try:
        P = _build_plan(state_dict, backbone, B, H, W, head_variant, num_classes, dense_heads, header_num_conv)
    finally:
        _CACHE = None
    P.cache = cache
    return P


PEAK_PATCH = 15          # 5 x 5 window of the first head conv x its three dilation-6 taps per axis


def build_peak_plan(state_dict, slots, map_hw, head_variant='rtm3d', cache=None, num_classes=3):
    """Patch plan of the peaks-only regression heads (csrc/sparse_heads.hip): one "image" per detection slot.
      zp  [slots, 15, 15, 256]  gathered samples of the fused map z (rtm3d_gather_peak_patches)
      h1p [slots, 5, 5, 2*256]  = ReLU(BN(conv3x3 dil 6)) of branches offset_fr_main | main_offset on the 5 x 5 window: on the
                                  patch layout the dilated conv is a conv with taps {0, 5, 10}^2 and no padding
      h2p [slots, 3, 3, 2*256]  = ReLU(BN(conv3x3)) on the 3 x 3 window (valid conv of the 5 x 5 one)
      out [slots, 16] and [slots, 2] fp32 = the two regression logits AT the peak (valid 3x3 conv of the 3 x 3 window)
    Window positions outside the map are zeroed after each conv: they are the next conv's zero padding (header.py:15-37).
    map_hw = (H/4, W/4) of the dense heat map."""
    global _CACHE
    if head_variant not in (None, 'rtm3d'):
        raise NotImplementedError('peaks-only regression heads exist for the rtm3d head table')
    _CACHE = cache
    try:
        heads = head_table(head_variant, num_classes)[1:3]          # offset_fr_main (16), main_offset (2); vertex_offset is never read
        P = Plan(slots, PEAK_PATCH, PEAK_PATCH)
        P.map_hw = (int(map_hw[0]), int(map_hw[1]))
        oc, G = 256, len(heads)
        zp = P.tensor(PEAK_PATCH, PEAK_PATCH, oc, 0, name='zp')
        h1 = P.tensor(5, 5, G * oc, 0, name='h1p')
        h2 = P.tensor(3, 3, G * oc, 0, name='h2p')
        ws, bs = zip(*[fold_bn(state_dict, 'detect_header.%s.0' % seq, 'detect_header.%s.1' % seq) for seq, _, _ in heads])
        w6 = np.concatenate(ws, 0).reshape(G * oc, oc, 9)
        P.conv_taps([zp], [h1], [w6], [np.concatenate(bs, 0)], [(5 * ky, 5 * kx) for ky in range(3) for kx in range(3)], 5, 5,
                    relu=True, name='peaks.conv_d6')
        P.patch_mask(h1, 2, name='peaks.mask_h1')
        taps3 = [(ky, kx) for ky in range(3) for kx in range(3)]
        ws, bs = zip(*[fold_bn(state_dict, 'detect_header.%s.3' % seq, 'detect_header.%s.4' % seq) for seq, _, _ in heads])
        P.conv_taps([P.sub(h1, g * oc, oc) for g in range(G)], [P.sub(h2, g * oc, oc) for g in range(G)],
                    [w.reshape(oc, oc, 9) for w in ws], bs, taps3, 3, 3, relu=True, name='peaks.conv_d1')
        P.patch_mask(h2, 1, name='peaks.mask_h2')
        for g, (seq, last, c) in enumerate(heads):
            w, b = fold_bn(state_dict, 'detect_header.%s.%s' % (seq, last))
            P.conv_taps([P.sub(h2, g * oc, oc)], [None], [w.reshape(c, oc, 9)], [b], taps3, 1, 1, name='peaks.out_%s' % seq,
                        out_nchw=g + 1)
        P.head_channels = [c for _, _, c in heads]
    finally:
        _CACHE = None
    P.cache = cache
    return P


def _build_plan(state_dict, backbone, B, H, W, head_variant, num_classes=3, dense_heads=None, header_num_conv=2):
    kind, depth = parse_backbone(backbone)
    if H % 32 or W % 32:
        raise ValueError('input height/width must be multiples of 32, got %dx%d' % (H, W))
    sd = state_dict
    P = Plan(B, H, W)
    oc = 256
    fch = [64, 128, 256, 512]
    fh = [(H // s, W // s) for s in (4, 8, 16, 32)]
    # neck concat buffers [up(256) | backbone feature]; the backbone writes its slice directly
    ncat = [P.tensor(fh[i][0], fh[i][1], oc + fch[i], 1) for i in range(3)]
    feat = [P.sub(ncat[i], oc, fch[i], name='feat%d' % i) for i in range(3)]
    feat.append(P.tensor(fh[3][0], fh[3][1], fch[3], 0, name='feat3'))
    if kind == 'dla':
        _build_dla(P, sd, H, W, feat)
    else:
        _build_resnet(P, sd, H, W, depth, feat)

    # ---- neck (models/nets/keypoint_fpn_fusion.py:35-46)
    # x[i-1] = proj_i(cat[up_i(head_i(x[i])), x[i-1]]) is only ever consumed by the next 1x1 `head` conv and
    # there is no non-linearity in between, so proj followed by head is ONE linear map: the two 1x1 convs
    # are composed on the host (fp64) into a single (256+C) -> 256 conv.  Same function, one rounding less,
    # and the C-channel intermediate never touches HBM (these 1x1 layers are bandwidth-bound).
    def compose(proj_key, head_key):
        if _CACHE is not None:
            return _CACHE.get('compose:%s|%s' % (proj_key, head_key), lambda: _compose(proj_key, head_key))
        return _compose(proj_key, head_key)

    def _compose(proj_key, head_key):
        wp, bp = fold_bn(sd, proj_key)                      # (C, 256+C, 1, 1)
        wh, bh = fold_bn(sd, head_key)                      # (256, C, 1, 1)
        w2 = wh[:, :, 0, 0].astype(np.float64) @ wp[:, :, 0, 0].astype(np.float64)
        b2 = wh[:, :, 0, 0].astype(np.float64) @ bp.astype(np.float64) + bh.astype(np.float64)
        return w2.astype(np.float32)[:, :, None, None], b2.astype(np.float32)

    hs = [None] * 4
    hs[3] = P.tensor(fh[3][0], fh[3][1], oc, 1, name='kfpn_h5')
    w, b = fold_bn(sd, 'kfpn_fusion.kfpn_head5')
    P.conv(feat[3], hs[3], w, b, name='kfpn_head5')
    for i in (3, 2, 1):
        L = i + 2
        P.deconv(hs[i], P.sub(ncat[i - 1], 0, oc), _np(sd, 'kfpn_fusion.kfpn_up%d.conv_tran.weight' % L).astype(np.float32),
                 name='kfpn_up%d' % L)
        w, b = compose('kfpn_fusion.kfpn_proj%d' % L, 'kfpn_fusion.kfpn_head%d' % (L - 1))
        if i > 1:
            hs[i - 1] = P.tensor(fh[i - 1][0], fh[i - 1][1], oc, 1, name='kfpn_h%d' % (L - 1))
            P.conv(ncat[i - 1], hs[i - 1], w, b, name='kfpn_proj%d+head%d' % (L, L - 1))
            P.ops[-1]['ref_macs_per_px'] = (oc + fch[i - 1]) * fch[i - 1] + fch[i - 1] * oc
    z0 = P.tensor(fh[0][0], fh[0][1], oc, 0, name='z0')
    P.conv(ncat[0], z0, w, b, name='kfpn_proj3+head2')
    P.ops[-1]['ref_macs_per_px'] = (oc + fch[0]) * fch[0] + fch[0] * oc
    # ---- fusion (keypoint_fpn_fusion.py:60-69): z = z0 + sum_i up^i(h_i) * softmax(up^i(h_i))
    us = []
    for i in (3, 2, 1):
        u = hs[i]
        for j in range(i):
            last = j == i - 1
            Hu, Wu = P.dims(u)
            nu = P.tensor(2 * Hu, 2 * Wu, oc, 0 if last else 1, name=('u%d' % (i + 2)) if last else None)
            P.deconv(u, nu, _np(sd, 'kfpn_fusion.fusion_up%d.%d.conv_tran.weight' % (i + 2, j)).astype(np.float32),
                     name='fusion_up%d.%d' % (i + 2, j))
            u = nu
        us.append(u)
    z = P.tensor(fh[0][0], fh[0][1], oc, 6, name='z')
    P.softmax_fuse(z0, z, us, name='kfpn_softmax_fuse')

    # ---- heads (models/nets/header.py:13-46): the d6 convs of all G branches fused into one 256->G*256 conv
    heads = head_table(head_variant, num_classes)
    if dense_heads is not None:
        heads = heads[:int(dense_heads)]
    G = len(heads)
    # (op names key the packed-weight cache: a plan with fewer dense branches packs other arrays under other names)
    hn = 'heads' if dense_heads is None else 'heads[:%d]' % G
    # MODEL.HEADER_NUM_CONV (header.py:12-13): one dilation-6 conv, then HEADER_NUM_CONV - 1 dilation-1 convs per branch (the shipped
    # configs: 2), as Sequential indices 3k / 3k + 1 of make_conv_level (utils/torch_utils.py:179-204).  The d1 convs ping-pong
    # between the two 4 x 256-channel tensors.
    nconv = int(header_num_conv)
    if nconv < 1:
        raise ValueError('MODEL.HEADER_NUM_CONV must be >= 1, got %r' % (header_num_conv,))
    h1 = P.tensor(fh[0][0], fh[0][1], G * oc, 1, name='h1')
    h2 = P.tensor(fh[0][0], fh[0][1], G * oc, 1, name='h2') if nconv > 1 else None
    ws, bs = zip(*[fold_bn(sd, 'detect_header.%s.0' % seq, 'detect_header.%s.1' % seq) for seq, _, _ in heads])
    P.conv(z, h1, np.concatenate(ws, 0), np.concatenate(bs, 0), dil=6, relu=True, name=hn + '.conv_d6')
    src, dst = h1, h2
    for k in range(1, nconv):
        ws, bs = zip(*[fold_bn(sd, 'detect_header.%s.%d' % (seq, 3 * k), 'detect_header.%s.%d' % (seq, 3 * k + 1)) for seq, _, _ in heads])
        P.grouped_conv([P.sub(src, g * oc, oc) for g in range(G)], [P.sub(dst, g * oc, oc) for g in range(G)], ws, bs,
                       relu=True, name=hn + ('.conv_d1' if k == 1 else '.conv_d1_%d' % k))
        src, dst = dst, src
    ws, bs = zip(*[fold_bn(sd, 'detect_header.%s.%s' % (seq, last)) for seq, last, _ in heads])
    P.headout(src, ws, bs, name=hn + '.out_convs')
    P.head_channels = [c for _, _, c in heads]
    return P


# ------------------------------------------------------------------------------------------ realize
def choose_bn_tile(cout, M):
    """Output-channel tile of the MFMA kernel (conv_mfma.hip).  128 channels go with the 256-pixel ring kernel (one
    8-wave workgroup per CU): taken when the layer still gives about a full round of 256 CUs; otherwise 64-channel tiles
    of the 128-pixel kernel (2-3 workgroups per CU) so that small late-stage layers fill the chip."""
    if cout < 32:
        return 16
    if cout < 64:
        return 32
    if cout % 128 == 0 and ((M + 255) // 256) * (cout // 128) >= 192:
        return 128
    return 64


V2_MIN_TILES_SHALLOW = 512   # layers with a short K loop (1x1 convs) or stride 2: two full rounds, as before


def choose_variant(cin, cout, M, groups, out_nchw, ntaps=9, stride=1):
    """2 = 256x256-tile 8-wave kernel (conv_mfma256.hip) when the layer has enough tiles to fill the
    chip at one workgroup per CU; 0 = 128-pixel-tile kernel (conv_mfma.hip); 3 = register-direct MFMA kernel for the
    4/16/32-channel layers of the stems (conv_smallc.hip)."""
    if cin % 64:
        if cin not in (4, 16, 32):
            raise NotImplementedError('no HIP kernel for a convolution with %d input channels (supported: 4, 16, 32, multiples of 64)' % cin)
        return 3
    if not out_nchw and cout % 256 == 0:
        # one persistent workgroup per CU (256 CUs): at least ~a round of tiles.  History: up to the end of round 2 the bar was two
        # full rounds (512): a single nearly full round - DLA level4, 240 tiles - was 12 % faster per launch on this kernel but
        # 0.1-0.5 ms SLOWER per pipelined step, because the previous batch's 3D decode (178 VGPRs x 8 waves and 67 KB of LDS per
        # workgroup, resident for ~3 ms) held ~60 CUs that a 160 KB workgroup cannot share.  With the direct-form solver
        # (116 VGPRs, 20 KB, ~1 ms beside the forward) the same switch gains 0.08-0.13 ms per step (round-3 same-box A/B:
        # 14.03 / 14.09 / 14.08 -> 13.95 / 13.96 / 14.01 ms).
        # (the 1x1 projections / roots and the stride-2 entry conv of level4 measured 0.023 / 0.052 ms on this kernel against
        # 0.019 / 0.049 on the 128-pixel one: the single-round bar is for stride-1 layers with at least 8 K-steps)
        deep = stride == 1 and ntaps * cin // 64 >= 8
        if ((M + 255) // 256) * (cout // 256) * groups >= (V2_MIN_TILES if deep else max(V2_MIN_TILES, V2_MIN_TILES_SHALLOW)):
            return 2
    return 0


def pack_mfma_weights(wt, bn):
    """wt: (taps, cout, cin) fp32 -> fp16 [ntile][kstep=(tap, cin/64)][bn rows][8 chunks][8] with the
    LDS bank swizzle (chunk ^= row & 7) baked in (see conv_mfma.hip)."""
    taps, cout, cin = wt.shape
    cpt = cin // 64
    cout_pad = (cout + bn - 1) // bn * bn
    w = np.zeros((taps, cout_pad, cin), np.float32)
    w[:, :cout] = wt
    w = w.reshape(taps, cout_pad // bn, bn, cpt, 8, 8)              # t, nt, r, q, c, e
    w = w.transpose(1, 0, 3, 2, 4, 5)                               # nt, t, q, r, c, e
    r = np.arange(bn)[:, None]
    cs = np.arange(8)[None, :]
    src = cs ^ (r & 7)                                              # chunk stored at position cs of row r
    w = w[:, :, :, r, src, :]
    return np.ascontiguousarray(w).astype(np.float16).reshape(-1), cout_pad


def pack_smallc_weights(wt, rows=False):
    """wt: (taps, cout, cin) fp32 -> fp16 [cout/16][k-step][lane = fk*16 + row][8] for conv_smallc.hip.
    cin=16: k-step s = taps (2s, 2s+1), or with rows=True (stride-1 layers, vertical-walk kernel) two
    k-steps per filter row: taps (ky,0),(ky,1) and (ky,2),zero; cin=32: k-step = tap; cin=4: k-step =
    filter row of 7 taps (+1 zero tap), lane group fk = taps (2fk, 2fk+1), 4 channels each."""
    taps, cout, cin = wt.shape
    if cin == 16 and rows:
        out = np.zeros((cout // 16, 6, 4, 16, 8), np.float32)          # ct, s = ky*2 + half, fk, row, j
        w = wt.reshape(taps, cout // 16, 16, cin)
        for ky in range(3):
            for fk in range(4):
                out[:, ky * 2, fk] = w[ky * 3 + (fk >> 1)][:, :, (fk & 1) * 8:(fk & 1) * 8 + 8]
                if fk < 2:
                    out[:, ky * 2 + 1, fk] = w[ky * 3 + 2][:, :, (fk & 1) * 8:(fk & 1) * 8 + 8]
        return np.ascontiguousarray(out).astype(np.float16).reshape(-1)
    S = 5 if cin == 16 else (taps if cin == 32 else 7)
    out = np.zeros((cout // 16, S, 4, 16, 8), np.float32)              # ct, s, fk, row, j
    w = wt.reshape(taps, cout // 16, 16, cin)
    for s in range(S):
        for fk in range(4):
            if cin == 16:
                t = 2 * s + (fk >> 1)
                if t < taps:
                    out[:, s, fk] = w[t][:, :, (fk & 1) * 8:(fk & 1) * 8 + 8]
            elif cin == 32:
                out[:, s, fk] = w[s][:, :, fk * 8:fk * 8 + 8]
            else:
                for half in range(2):
                    kx = 2 * fk + half
                    if kx < 7:
                        out[:, s, fk, :, half * 4:half * 4 + 4] = w[s * 7 + kx]
    return np.ascontiguousarray(out).astype(np.float16).reshape(-1)


def conv64_eligible(op):
    """3x3 / stride 1 / dilation 1 / 64 -> 64 channels on a map that 8 x 32 pixel tiles cover: conv64_halo.hip."""
    taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    return (op['cin'] == 64 and op['cout'] == 64 and op['groups'] == 1 and op['in_stride'] == 1 and op['out_scale'] == 1
            and not op['out_nchw'] and list(op['taps'][0]) == taps3 and op['Hm'] % 8 == 0 and op['Wm'] % 32 == 0)


BN_TILE_OVERRIDE = {}     # DIAGNOSTIC (bench.py --bn-tile): op name -> output-channel tile of the 128-pixel kernel, for A/B runs
USE_CONV128 = True
USE_CONV64S2 = os.environ.get('RTM3D_CONV64S2', '1') != '0'   # 64 -> 128 stride-2 entry convs on conv64s2_halo.hip (A/B switch)
C128_MIN_TILES = 256     # at least one 8 x 32-pixel tile per CU, else the small-launch kernels of conv_mfma.hip do better (ResNet-18 bs=8, 240 tiles: 3.87 ms per step on this kernel against 3.76)


def conv128_eligible(op, B):
    """3x3 / stride 1 / dilation 1 / 128 -> 128 channels on a map that 8 x 32 pixel tiles cover, with at least one tile per
    CU: conv128_halo.hip (DLA-34 level3, ResNet layer2).  (The kernel takes any multiples of 128 channels; the plan only
    sends it the layers it was measured to win on.)"""
    taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    return (op['cin'] == 128 and op['cout'] == 128 and op['groups'] == 1 and op['in_stride'] == 1 and op['out_scale'] == 1
            and not op['out_nchw'] and list(op['taps'][0]) == taps3 and op['Hm'] % 8 == 0 and op['Wm'] % 32 == 0
            and B * (op['Hm'] // 8) * (op['Wm'] // 32) >= C128_MIN_TILES)


def conv64s2_eligible(op):
    """3x3 / STRIDE 2 / dilation 1 / 64 -> 128 channels onto a map that 4 x 32 pixel tiles cover: conv64s2_halo.hip."""
    taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    return (USE_CONV64S2 and op['cin'] == 64 and op['cout'] == 128 and op['groups'] == 1 and op['in_stride'] == 2 and op['out_scale'] == 1
            and not op['out_nchw'] and op['res'][0] is None and list(op['taps'][0]) == taps3 and op['Hm'] % 4 == 0 and op['Wm'] % 32 == 0
            and 'tap_dc' not in op)


def pack_conv64s2_weights(wt):
    """wt: (9, 128, 64) fp32 [tap][cout][cin] -> fp16 [tap][k half][16-channel tile of 8][lane = fk*16 + row][8] (conv64s2_halo.hip)."""
    w = wt.reshape(9, 8, 16, 2, 4, 8)                 # tap, ct, row, kk, fk, j
    w = w.transpose(0, 3, 1, 4, 2, 5)                 # tap, kk, ct, fk, row, j
    return np.ascontiguousarray(w).astype(np.float16).reshape(-1)


def pack_conv64_weights(wt):
    """wt: (9, 64, 64) fp32 [tap][cout][cin] -> fp16 [tap][k half][16-channel tile][lane = fk*16 + row][8]: the MFMA A
    fragments a wave of conv64_halo.hip keeps in registers."""
    w = wt.reshape(9, 4, 16, 2, 4, 8)                 # tap, ct, row, kk, fk, j
    w = w.transpose(0, 3, 1, 4, 2, 5)                 # tap, kk, ct, fk, row, j
    return np.ascontiguousarray(w).astype(np.float16).reshape(-1)


S2D_ONLY = os.environ.get('RTM3D_S2D_ONLY', '1') != '0'   # a fused level tail skips the ordinary copy of its output when every reader can take the space-to-depth one


def folds_pre(rp):
    """(indices of convs that the project fold rewrites: they keep their ordinary input)"""
    return set(rp._project_folds().keys()) if FOLD_PROJECT else set()


def A_bias(J, D):
    """A_up applied to the transposed conv's own bias (zero for the reference's UpSample, models/nets/module.py:9: bias=False)."""
    A_up = J['w'][0][0][:, :256].astype(np.float64)
    return A_up @ D['bias'][0].astype(np.float64)


def pack_root64_weights(wt):
    """wt: (64, 128) fp32 [cout][cat channel] (cat = [x2 | x1]) -> fp16 [4 output tiles][4 K-steps][lane = fk*16 + row][8] for
    conv64_root.hip: element j of lane (fk, row) in K-step s is cat channel s*32 + (j >> 2)*16 + fk*4 + (j & 3) - the order in
    which a lane's two accumulator fragments (4 consecutive channels of two 16-channel tiles) line up as an MFMA B operand."""
    out = np.zeros((4, 4, 4, 16, 8), np.float32)                       # ct, s, fk, row, j
    for s_ in range(4):
        for fk in range(4):
            for j in range(8):
                out[:, s_, fk, :, j] = wt[:, s_ * 32 + (j >> 2) * 16 + fk * 4 + (j & 3)].reshape(4, 16)
    return np.ascontiguousarray(out).astype(np.float16).reshape(-1)


def pack_headout_weights(ws, biases):
    """4 x (cout<=16, 256, 3, 3) -> fp16 [head][tap][chunk*2+kk][lane=fk*16+row][8], fp32 bias [head][16]."""
    G = len(ws)
    out = np.zeros((G, 9, 8, 4, 16, 8), np.float32)                    # head, tap, kblk, fk, row, j
    bias = np.zeros((G, 16), np.float32)
    for h, (w, b) in enumerate(zip(ws, biases)):
        co = w.shape[0]
        wt = w.reshape(co, 8, 4, 8, 9)                                  # cout, kblk(=cin/32), fk, j, tap
        out[h, :, :, :, :co, :] = wt.transpose(4, 1, 2, 0, 3)
        bias[h, :co] = b
    return np.ascontiguousarray(out).astype(np.float16).reshape(-1), bias.reshape(-1)


class RealizedPlan(object):
    """A Plan recorded into a librtm3d_hip context."""
    def __init__(self, plan, device_index):
        lib = _lib.load()
        self.lib, self.plan = lib, plan
        self.cache = getattr(plan, 'cache', None)
        ctx = ctypes.c_void_p()
        _lib.check(lib.rtm3d_ctx_create(int(device_index), ctypes.byref(ctx)), 'ctx_create')
        self.ctx = ctx
        self._keep = []
        self.tids = []
        self._stat_slots = self._softmax_stat_producers()
        tail = self._level_tail_chains() if FUSE_LEVEL_TAIL else {}
        nfold = self._neck_up_folds(tail) if FOLD_NECK_UP else []
        widen = {f['hs'].tid: 4 * f['Cf'] for f in nfold}    # device tensors that also hold a space-to-depth copy of a backbone feature
        self._s2d_for = {f['feat']: (f['hs'], plan.tensors[f['hs'].tid]['C']) for f in nfold if 'feat' in f}
        self._plan_s2d_only(nfold, tail)
        for i, t in enumerate(plan.tensors):
            tid = ctypes.c_int()
            _lib.check(lib.rtm3d_tensor_create(ctx, plan.B, t['H'], t['W'], t['C'] + widen.get(i, 0), t['pad'], ctypes.byref(tid)), 'tensor_create')
            self.tids.append(tid.value)
        self.op_names = []                      # one entry per RECORDED runtime op (a fused pair records one)
        self.weight_ranges = []                 # per recorded conv: largest |weight| / |bias| as realized (after the level rewrites)
        fused = self._stem_fusion_pairs() if FUSE_STEM else {}
        entry = self._level_entry_triples() if FUSE_LEVEL_ENTRY else {}
        folds = self._project_folds() if FOLD_PROJECT else {}
        skip = set(folds.values()) | {f['up'] for f in nfold}
        s2d_of = {f['tail']: f for f in nfold if 'tail' in f}   # conv64_root launch -> the fold its second (space-to-depth) output feeds
        neck_by = {f['pj']: f for f in nfold}
        # output slices of the un-rewritten plan that this realization never writes (for verify.range_report): the `up` maps the
        # neck fold composes away, the projected residuals, feature maps that exist only as their space-to-depth copy
        self.unwritten = set()
        for kk in skip:
            for o in plan.ops[kk]['out'] if plan.ops[kk]['op'] == 'conv' else []:
                if o is not None:
                    self.unwritten.add((o.tid, o.coff, o.C))
        for kk in getattr(self, '_s2d_only_producers', ()):
            o = plan.ops[kk]['out'][0]
            self.unwritten.add((o.tid, o.coff, o.C))
        for kk, f in s2d_of.items():
            if f.get('s2d_only'):
                o = plan.ops[tail[kk][0]]['out'][0]
                self.unwritten.add((o.tid, o.coff, o.C))
        folded_by = {}
        for c2k, pjk in folds.items():
            folded_by[c2k] = self._folded_conv(plan.ops[c2k], plan.ops[pjk])
        for k, op in enumerate(plan.ops):
            self._k = k
            if k in skip:
                continue
            if k in fused:
                chain = [plan.ops[j] for j in fused[k]]
                self._op_stem_fused(op, *chain)
                self.op_names.append('+'.join([op['name']] + [c['name'].split('.')[-1] for c in chain]))
                skip.update(fused[k])
                continue
            if k in entry:
                proj, conv = plan.ops[entry[k][0]], plan.ops[entry[k][1]]
                self._op_conv32s2_fused(op, proj, conv)
                self.op_names.append(op['name'] + '+project+' + conv['name'].split('.', 2)[-1])
                skip.update(entry[k])
                continue
            if k in tail:
                root = plan.ops[tail[k][0]]
                pool = plan.ops[tail[k][1]] if len(tail[k]) > 1 else None
                f = s2d_of.get(k)
                self._op_conv64_root(op, root, pool, s2d=(f['hs'], plan.tensors[f['hs'].tid]['C']) if f else None,
                                     skip_out=bool(f and f.get('s2d_only')))
                self.op_names.append(op['name'] + '+root' + ('+' + pool['name'].split('.', 1)[-1] if pool else '') + ('+s2d' if f else ''))
                skip.update(tail[k])
                continue
            if k in folded_by:
                op = folded_by[k]
            if k in self._s2d_readers:
                kind, hs_, base_, cf_ = self._s2d_readers[k]
                if kind == 'pool':
                    _lib.check(self.lib.rtm3d_op_maxpool_s2d(self.ctx, self.tids[hs_.tid], base_, self.tids[op['out'].tid], op['out'].coff, cf_), 'op_maxpool_s2d ' + op['name'])
                    self.op_names.append(op['name'])
                    continue
                op = self._s2d_input_conv(op, hs_, base_, cf_)
            if k in neck_by:
                op = self._neck_fold_conv(neck_by[k])
            getattr(self, '_op_' + op['op'])(op)
            self.op_names.append(op['name'])

    def _plan_s2d_only(self, nfold, tail):
        """Which feature maps exist ONLY as their space-to-depth copy, and how their remaining readers take it (sets self._in_s2d_for,
        self._s2d_only, self._s2d_readers, self._s2d_only_producers; marks the folds)."""
        plan = self.plan
        # a level tail whose ordinary output has, besides the folded 1x1 and its own pool, ONE reader that can take the space-to-depth
        # copy instead (the next level's stride-2 entry on conv64s2_halo.hip) does not write the ordinary copy at all
        self._in_s2d_for, self._s2d_only = {}, {}
        for f in nfold:
            if 'tail' not in f or not S2D_ONLY:
                continue
            ro = plan.ops[tail[f['tail']][0]]['out'][0]
            own = set([f['tail'], f['up'], f['pj']] + list(tail[f['tail']]))
            readers = [j for j, o in enumerate(plan.ops) if j not in own and self._reads(o, ro.tid) and self._reads_slice(o, ro)]
            if (len(readers) == 1 and conv64s2_eligible(plan.ops[readers[0]]) and plan.ops[readers[0]].get('variant') is None
                    and plan.B * (plan.ops[readers[0]]['Hm'] // 4) * (plan.ops[readers[0]]['Wm'] // 32) >= 64
                    and plan.ops[readers[0]]['inp'][0].coff == ro.coff and readers[0] not in folds_pre(self)):
                f['s2d_only'] = True
                self._in_s2d_for[readers[0]] = (f['hs'], plan.tensors[f['hs'].tid]['C'])
                self._s2d_only[(ro.tid, ro.coff, ro.C)] = (f['hs'], plan.tensors[f['hs'].tid]['C'])
        # likewise a plain conv on the 128-pixel kernel (DLA level3 / level4 roots): every other reader of its output must be a 2x2 / 2
        # max-pool (-> rtm3d_op_maxpool_s2d) or a stride-2 conv with taps within +-1 (-> the same conv restated on the copy)
        self._s2d_readers, self._s2d_only_producers = {}, set()
        for f in nfold:
            if 'feat' not in f or not S2D_ONLY or conv64_eligible(plan.ops[f['feat']]):
                continue
            ro = plan.ops[f['feat']]['out'][0]
            own = {f['feat'], f['up'], f['pj']}
            readers = [j for j, o in enumerate(plan.ops) if j not in own and self._reads(o, ro.tid) and self._reads_slice(o, ro)]
            plan_ = {}
            for j in readers:
                o = plan.ops[j]
                same = lambda sl: sl.tid == ro.tid and sl.coff == ro.coff and sl.C == ro.C
                if o['op'] == 'maxpool' and o['k'] == 2 and o['stride'] == 2 and o['pad'] == 0 and same(o['inp']):
                    plan_[j] = 'pool'
                elif (o['op'] == 'conv' and o['groups'] == 1 and o['in_stride'] == 2 and o['out_scale'] == 1 and same(o['inp'][0]) and o['cin'] % 64 == 0
                      and all(abs(dy) <= 1 and abs(dx) <= 1 for dy, dx in o['taps'][0]) and o.get('variant') is None and 'tap_dc' not in o
                      and (o['res'][0] is None or o['res'][0].tid != ro.tid) and j not in folds_pre(self)
                      and len(o['taps'][0]) * (o['cin'] // 64) <= _lib.MAX_TAPS and not conv64s2_eligible(o)):
                    plan_[j] = 'conv'
                else:
                    plan_ = None
                    break
            if plan_ is None:
                continue
            f['s2d_only'] = True
            self._s2d_only_producers.add(f['feat'])
            base = plan.tensors[f['hs'].tid]['C']
            for j, kind in plan_.items():
                self._s2d_readers[j] = (kind, f['hs'], base, f['Cf'])
            self._s2d_only[(ro.tid, ro.coff, ro.C)] = (f['hs'], base)

    @classmethod
    def rewrites_only(cls, plan):
        """An object that can find and build the realize-level rewrites of `plan` (project folds, level tails, neck up-folds)
        WITHOUT a device context: for the CPU tests of their algebra and tap tables (tests/test_plan_cpu.py)."""
        self = cls.__new__(cls)
        self.plan, self.cache, self.ctx = plan, getattr(plan, 'cache', None), None
        self._stat_slots = self._softmax_stat_producers()
        return self

    def _neck_up_folds(self, tail):
        """Neck (models/nets/keypoint_fpn_fusion.py:35-46): x[i-1] = head(proj(cat[up(h), feat])) with up = ConvTranspose2d without
        bias or non-linearity and proj∘head already composed into ONE 1x1 conv A = [A_up | A_f]:
            out = A_up (up h) + A_f feat + b  =  up'(h) + A_f feat + b,     up' = the transposed conv with weights A_up W (per tap).
        Run as ONE launch over the transposed conv's input grid, the 1x1's share of the FEATURE map is Cf / 64 more K-steps per
        sub-pixel phase - provided the feature pixel (2y + py, 2x + px) can be addressed from grid position (y, x): the feature's
        producer writes a second, space-to-depth copy into spare channels of the tensor that holds h (rtm3d_conv_desc.tap_dc names
        them).  The `up` map (0.5 GB at bs=32 for the last level) is never written or read, one launch per level goes away.  (Round 2
        tried the same algebra with the feature term as an epilogue residual of the transposed conv: slower.  Here it is K-steps.)
        Matches (deconv D onto channels [0, 256) of T; 1x1 J over all of T = [up | feat]) where feat is written by the root of a
        fused level tail (conv64_root.hip) or by a plain conv that can run on the 128-pixel kernel (both can emit the copy).
        Returns [{'up', 'pj', 'hs', 'Cf', 'tail' | 'feat'}]."""
        P = self.plan
        out = []
        for kj, J in enumerate(P.ops):
            if (J['op'] != 'conv' or J['groups'] != 1 or list(J['taps'][0]) != [(0, 0)] or J['in_stride'] != 1 or J['out_scale'] != 1 or J['relu']
                    or J['res'][0] is not None or J['out_nchw'] or J['cout'] != 256 or J['cin'] <= 256 or (J['cin'] - 256) % 64 or J.get('variant') is not None):
                continue
            T, Cf = J['inp'][0], J['cin'] - 256
            if T.coff != 0 or P.tensors[T.tid]['C'] != J['cin']:
                continue
            ups = [k for k in range(kj) if P.ops[k]['op'] == 'conv' and P.ops[k]['groups'] == 4 and P.ops[k]['out_scale'] == 2
                   and all(o is not None and o.tid == T.tid and o.coff == 0 and o.C == 256 for o in P.ops[k]['out'])]
            if len(ups) != 1:
                continue
            ku = ups[0]
            D = P.ops[ku]
            hs = D['inp'][0]
            ok = (D['cin'] == 256 and D['cout'] == 256 and all(len(tp) == 4 for tp in D['taps']) and not D['relu'] and all(r is None for r in D['res'])
                  and all(i.tid == hs.tid and i.coff == 0 for i in D['inp']) and hs.C == 256 and P.tensors[hs.tid]['C'] == 256
                  and P.tensors[hs.tid]['pad'] >= 1 and self._stat_slots_of(ku) < 0 and D.get('variant') is None
                  and (2 * D['Hm'], 2 * D['Wm']) == P.dims(J['out'][0]) and J['out'][0].tid not in (T.tid, hs.tid)
                  and ((P.B * D['Hm'] * D['Wm'] + 255) // 256) * 4 >= V2_MIN_TILES and 16 + Cf // 64 <= _lib.MAX_TAPS)
            # nothing but J reads the `up` slice
            readers = [j for j, o in enumerate(P.ops) if j != ku and self._reads_slice(o, Slice(T.tid, 0, 256))]
            ok = ok and readers == [kj]
            # who writes the feature slice [256, 256 + Cf)
            fold = {'up': ku, 'pj': kj, 'hs': hs, 'Cf': Cf}
            if ok:
                kt = [k for k, ch in tail.items() if (lambda o: o.tid == T.tid and o.coff == 256 and o.C == Cf)(P.ops[ch[0]]['out'][0])]
                kf = [k for k, o in enumerate(P.ops) if o['op'] == 'conv' and o['groups'] == 1 and o['out_scale'] == 1 and not o['out_nchw'] and o['cin'] % 64 == 0
                      and o['out'][0] is not None and o['out'][0].tid == T.tid and o['out'][0].coff == 256 and o['out'][0].C == Cf
                      and o.get('variant') is None and o['Hm'] % 2 == 0 and o['Wm'] % 2 == 0 and o['cout'] % 16 == 0
                      and not any(k in ch for ch in tail.values()) and k not in tail]
                if len(kt) == 1 and Cf == 64 and kt[0] < ku:
                    fold['tail'] = kt[0]
                elif len(kf) == 1 and kf[0] < ku and not conv64s2_eligible(P.ops[kf[0]]):
                    fold['feat'] = kf[0]            # (the 128-pixel kernel, or conv64_halo.hip for a 64 -> 64 3x3: both can emit the copy)
                else:
                    ok = False
            # between the deconv's place and the 1x1's nothing may write h (the fused op runs at the 1x1's place)
            if ok:
                for j in range(ku + 1, kj):
                    o = P.ops[j]
                    outs = o['out'] if o['op'] == 'conv' else [o.get('out')] if o['op'] == 'maxpool' else [o.get('z_out')] if o['op'] == 'softmax' else []
                    if any(t is not None and t.tid == hs.tid for t in outs):
                        ok = False
            if ok:
                out.append(fold)
        return out

    def _stat_slots_of(self, k):
        return self._stat_slots.get(k, -1)

    def _neck_fold_conv(self, f):
        P = self.plan
        D, J, hs = P.ops[f['up']], P.ops[f['pj']], f['hs']
        Cf, e = f['Cf'], f['Cf'] // 64

        def make():
            A = J['w'][0][0].astype(np.float64)                        # (256, 256 + Cf) = [A_up | A_f]
            A_up, A_f = A[:, :256], A[:, 256:]
            ws = []
            for g in range(4):
                wt = np.zeros((4 * 4 + e, 256, 64), np.float64)
                for t in range(4):
                    Wc = A_up @ D['w'][g][t].astype(np.float64)        # (cout, cin) of the composed tap
                    for q in range(4):
                        wt[t * 4 + q] = Wc[:, q * 64:(q + 1) * 64]
                for q in range(e):
                    wt[16 + q] = A_f[:, q * 64:(q + 1) * 64]
                ws.append(wt)
            return np.stack(ws, 0).astype(np.float32)
        w = self.cache.get('neckfold:%s|%s' % (D['name'], J['name']), make) if self.cache is not None else make()
        taps, dcs = [], []
        base = P.tensors[hs.tid]['C']                                   # the space-to-depth copy sits behind h's own channels
        for g in range(4):
            py, px = D['out_off'][g]
            tl = [(dy, dx) for (dy, dx) in D['taps'][g] for _ in range(4)] + [(0, 0)] * e
            dc = [q * 64 for _ in range(4) for q in range(4)] + [base + (py * 2 + px) * Cf + q * 64 for q in range(e)]
            taps.append(tl); dcs.append(dc)
        bias = J['bias'][0].astype(np.float64) + A_bias(J, D)
        return {'op': 'conv', 'name': D['name'] + '+' + J['name'], 'inp': [hs] * 4, 'out': [J['out'][0]] * 4, 'res': [None] * 4,
                'Hm': D['Hm'], 'Wm': D['Wm'], 'in_stride': 1, 'out_scale': 2, 'cin': 64, 'cout': 256, 'groups': 4, 'taps': taps, 'tap_dc': dcs,
                'out_off': list(D['out_off']), 'relu': False, 'w': w, 'bias': np.tile(bias.astype(np.float32)[None], (4, 1)), 'out_nchw': 0,
                'out_hw': (2 * D['Hm'], 2 * D['Wm'])}

    @staticmethod
    def _s2d_input_conv(op, hs, base, Cf):
        """A stride-2 conv (taps within +-1) restated on the space-to-depth copy of its input - full-resolution pixel (2y + d, 2x + e)
        is half-resolution pixel (y + d // 2, x + e // 2), phase (d % 2, e % 2) - as a STRIDE-1 conv over 64-channel pseudo-taps
        (tap_dc): same products, same order within a tap."""
        cout, cpt = op['cout'], op['cin'] // 64
        taps, w = op['taps'][0], op['w'][0]                                  # (taps, cout, cin)
        wt = np.zeros((len(taps) * cpt, cout, 64), np.float32)
        tl, dc = [], []
        for t, (dy, dx) in enumerate(taps):
            for q in range(cpt):
                wt[t * cpt + q] = w[t][:, q * 64:(q + 1) * 64]
                tl.append((dy // 2, dx // 2)); dc.append(base + ((dy % 2) * 2 + (dx % 2)) * Cf + q * 64)
        new = dict(op)
        new.update(cin=64, in_stride=1, taps=[tl], tap_dc=[dc], w=wt[None], inp=[Slice(hs.tid, 0, 64)], name=op['name'] + '[s2d]')
        return new

    def _project_folds(self):
        """{index of a block's second conv: index of the `project` 1x1 that produces its residual} where the 1x1 (no ReLU, stride
        1, a multiple of 64 input channels) reads another channel slice of the tensor the conv reads, and nothing else reads its
        output: conv(t) + bias + project(b) is ONE convolution over [t | b] whose extra taps are the centre pixel of b
        (models/nets/dla.py:92-99,190-198: out = bn2(conv2(.)) + project(bottom), then ReLU).  Exact in real arithmetic; in fp16
        storage it saves the rounding of the projected map.  The 1x1 launch, its output tensor and the residual read go away."""
        P = self.plan
        out = {}
        for k, c2 in enumerate(P.ops):
            if (c2['op'] != 'conv' or c2['groups'] != 1 or c2['res'][0] is None or c2['in_stride'] != 1 or c2['out_scale'] != 1
                    or c2['out_nchw'] or c2['cin'] % 64 or c2.get('variant') is not None or 'tap_dc' in c2):
                continue
            R = c2['res'][0]
            prods = [j for j in range(k) if P.ops[j]['op'] == 'conv' and any(o is not None and o.tid == R.tid for o in P.ops[j]['out'])]
            if len(prods) != 1:
                continue
            pj = P.ops[prods[0]]
            users = [j for j, o in enumerate(P.ops) if j != prods[0] and self._reads(o, R.tid)]
            ok = (pj['groups'] == 1 and list(pj['taps'][0]) == [(0, 0)] and pj['in_stride'] == 1 and pj['out_scale'] == 1 and not pj['relu']
                  and pj['res'][0] is None and not pj['out_nchw'] and pj['cin'] % 64 == 0 and pj['cout'] == c2['cout']
                  and pj['out'][0].coff == R.coff and pj['out'][0].C == R.C and pj.get('variant') is None and users == [k]
                  and pj['inp'][0].tid == c2['inp'][0].tid and (pj['Hm'], pj['Wm']) == (c2['Hm'], c2['Wm'])
                  and len(c2['taps'][0]) * (c2['cin'] // 64) + pj['cin'] // 64 <= _lib.MAX_TAPS
                  and not any(n.tid == R.tid for n in P.named.values()))
            if ok and not FOLD_PROJECT_C128 and USE_CONV128 and conv128_eligible(c2, P.B):
                ok = False
            if ok and conv64_eligible(c2):
                ok = False                      # (the register-resident 64-channel kernel has no generic taps)
            if ok:
                out[k] = prods[0]
        return out

    @staticmethod
    def _folded_conv(c2, pj):
        cout, cpt, e = c2['cout'], c2['cin'] // 64, pj['cin'] // 64
        taps, w2, wp = c2['taps'][0], c2['w'][0], pj['w'][0][0]              # w2: (taps, cout, cin), wp: (cout, cin_b)
        wt = np.zeros((len(taps) * cpt + e, cout, 64), np.float32)
        tl, dc = [], []
        for t, (dy, dx) in enumerate(taps):
            for q in range(cpt):
                wt[t * cpt + q] = w2[t][:, q * 64:(q + 1) * 64]
                tl.append((dy, dx)); dc.append(q * 64)
        delta = pj['inp'][0].coff - c2['inp'][0].coff
        for q in range(e):
            wt[len(taps) * cpt + q] = wp[:, q * 64:(q + 1) * 64]
            tl.append((0, 0)); dc.append(delta + q * 64)
        op = dict(c2)
        op.update(cin=64, taps=[tl], tap_dc=[dc], w=wt[None], bias=c2['bias'] + pj['bias'], res=[None], name=c2['name'] + '+project')
        return op

    def _level_tail_chains(self):
        """{index of a 64 -> 64 3x3 conv with a residual (conv64_halo-eligible): [index of the 1x1 128 -> 64 conv over
        cat[that conv's output | its residual] (the tree's root; the conv's output has no other reader) (, index of a 2x2/2
        max-pool of the root's output, when it directly follows)]}: one launch (conv64_root.hip); x2 is never written."""
        P = self.plan
        out = {}
        for k, cv in enumerate(P.ops[:-1]):
            rt = P.ops[k + 1]
            if cv['op'] != 'conv' or not conv64_eligible(cv) or cv['res'][0] is None or cv.get('variant') is not None:
                continue
            x2, x1 = cv['out'][0], cv['res'][0]
            users = [j for j, o in enumerate(P.ops) if j != k and self._reads(o, x2.tid) and self._reads_slice(o, x2)]
            ok = (rt['op'] == 'conv' and rt['cin'] == 128 and rt['cout'] == 64 and rt['groups'] == 1 and list(rt['taps'][0]) == [(0, 0)]
                  and rt['in_stride'] == 1 and rt['out_scale'] == 1 and rt['res'][0] is None and not rt['out_nchw'] and rt.get('variant') is None
                  and x2.tid == x1.tid and x1.coff == x2.coff + 64 and rt['inp'][0].tid == x2.tid and rt['inp'][0].coff == x2.coff
                  and users == [k + 1] and rt['out'][0].coff % 8 == 0
                  and not (rt['out'][0].tid == x2.tid and rt['out'][0].coff < x2.coff + 128 and x2.coff < rt['out'][0].coff + 64)
                  and rt['out'][0].tid != cv['inp'][0].tid
                  and not any(n.tid == x2.tid and n.coff < x2.coff + 64 and x2.coff < n.coff + n.C for n in P.named.values()))
            if not ok:
                continue
            chain = [k + 1]
            if k + 2 < len(P.ops):
                m = P.ops[k + 2]
                ro = rt['out'][0]
                if (m['op'] == 'maxpool' and m['k'] == 2 and m['stride'] == 2 and m['pad'] == 0 and m['inp'].tid == ro.tid
                        and m['inp'].coff == ro.coff and m['inp'].C == 64 and m['out'].coff % 8 == 0 and m['out'].tid != ro.tid):
                    chain.append(k + 2)
            out[k] = chain
        return out

    @staticmethod
    def _reads_slice(op, s):
        """Does `op` read any channel of Slice s (same tensor)?"""
        def hit(t, C):
            return t is not None and t.tid == s.tid and t.coff < s.coff + s.C and s.coff < t.coff + C
        if op['op'] == 'conv':
            return any(hit(i, op['cin']) for i in op['inp']) or any(hit(r_, op['cout']) for r_ in op['res'])
        if op['op'] in ('maxpool', 'headout'):
            return hit(op['inp'], op['inp'].C)
        if op['op'] == 'patch_mask':
            return hit(op['t'], op['t'].C)
        if op['op'] == 'softmax':
            return hit(op['z_in'], op['z_in'].C) or any(hit(u, u.C) for u in op['us'])
        return False

    def _op_conv64_root(self, cv, rt, pool, s2d=None, skip_out=False):
        f32 = lambda v: self._blob(np.ascontiguousarray(v, np.float32))
        x, x1, ro = cv['inp'][0], cv['res'][0], rt['out'][0]
        po = pool['out'] if pool is not None else None
        wc = self._packed(cv, 0, 'c64', 0, lambda: pack_conv64_weights(cv['w'][0]))
        wr = self._packed(rt, 0, 'root64', 0, lambda: pack_root64_weights(rt['w'][0][0]))
        _lib.check(self.lib.rtm3d_op_conv64_root(self.ctx, self.tids[x.tid], x.coff, self.tids[x1.tid], x1.coff, 1 if cv['relu'] else 0,
                                                 self._blob(wc), f32(cv['bias'][0]), self._blob(wr), f32(rt['bias'][0]),
                                                 -1 if skip_out else self.tids[ro.tid], ro.coff, 1 if rt['relu'] else 0,
                                                 self.tids[po.tid] if po is not None else -1, po.coff if po is not None else 0,
                                                 self.tids[s2d[0].tid] if s2d is not None else -1, s2d[1] if s2d is not None else 0),
                   'op_conv64_root')

    def _stem_fusion_pairs(self):
        """{index of the 7x7 NHWC4 stem conv: [index of the 3x3 16->16 conv that is its only consumer (, index of the 3x3
        stride-2 16->32 conv that is THAT one's only consumer)]}: the chain becomes one launch (conv_stem_fused.hip) and
        the 16-channel full-resolution maps between the layers never touch HBM.  FUSE_STEM: True = as deep as possible,
        2 = the first two layers only (A/B, tests), False = off."""
        P = self.plan
        taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]

        def only_user(k, out):
            users = [j for j, o in enumerate(P.ops) if j != k and self._reads(o, out.tid)]
            return users[0] if len(users) == 1 and users[0] == k + 1 else None

        def plain3x3(b, src, cin, cout, stride):
            return (b['op'] == 'conv' and b['cin'] == cin and b['cout'] == cout and b['groups'] == 1 and b['in_stride'] == stride
                    and b['out_scale'] == 1 and list(b['taps'][0]) == taps3 and b['relu'] and b['res'][0] is None and not b['out_nchw']
                    and b['inp'][0].tid == src.tid and b['inp'][0].coff == src.coff)

        pairs = {}
        for k, a in enumerate(P.ops):
            if a['op'] != 'conv' or a['cin'] != 4 or a['cout'] != 16 or a['in_stride'] != 1 or len(a['taps'][0]) != 49 or not a['relu']:
                continue
            if a['Hm'] % 16 or a['Wm'] % 32 or P.tensors[a['inp'][0].tid]['pad'] < 4:
                continue
            j = only_user(k, a['out'][0])
            if j is None or not plain3x3(P.ops[j], a['out'][0], 16, 16, 1):
                continue
            chain = [j]
            j2 = only_user(j, P.ops[j]['out'][0])
            if FUSE_STEM is True and j2 is not None and plain3x3(P.ops[j2], P.ops[j]['out'][0], 16, 32, 2) and P.ops[j2]['out'][0].coff % 8 == 0:
                chain.append(j2)
            pairs[k] = chain
        return pairs

    def _level_entry_triples(self):
        """{index of a 2x2/2 max-pool of a 32-channel map: [index of the 1x1 32->64 conv on the pooled map (no ReLU),
        index of the 3x3 stride-2 32->64 conv (ReLU) on the SAME input]} when the pooled map has no other reader:
        one launch (conv32s2_fused.hip)."""
        P = self.plan
        taps3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
        out = {}
        for k, m in enumerate(P.ops[:-2]):
            if m['op'] != 'maxpool' or m['k'] != 2 or m['stride'] != 2 or m['pad'] != 0 or m['inp'].C != 32:
                continue
            pj, cv = P.ops[k + 1], P.ops[k + 2]
            users = [j for j, o in enumerate(P.ops) if j != k and self._reads(o, m['out'].tid)]
            Hi, Wi = P.dims(m['inp'])
            ok = (users == [k + 1] and pj['op'] == 'conv' and cv['op'] == 'conv'
                  and pj['cin'] == 32 and pj['cout'] == 64 and pj['groups'] == 1 and list(pj['taps'][0]) == [(0, 0)] and pj['in_stride'] == 1
                  and pj['out_scale'] == 1 and not pj['relu'] and pj['res'][0] is None and not pj['out_nchw']
                  and pj['inp'][0].tid == m['out'].tid and pj['inp'][0].coff == m['out'].coff
                  and cv['cin'] == 32 and cv['cout'] == 64 and cv['groups'] == 1 and list(cv['taps'][0]) == taps3 and cv['in_stride'] == 2
                  and cv['out_scale'] == 1 and cv['relu'] and cv['res'][0] is None and not cv['out_nchw']
                  and cv['inp'][0].tid == m['inp'].tid and cv['inp'][0].coff == m['inp'].coff
                  and Hi % 16 == 0 and Wi % 64 == 0 and P.tensors[m['inp'].tid]['pad'] >= 1
                  and pj['out'][0].coff % 8 == 0 and cv['out'][0].coff % 8 == 0)
            if ok:
                out[k] = [k + 1, k + 2]
        return out

    def _op_conv32s2_fused(self, pool, proj, conv):
        def pack_conv():
            w = conv['w'][0].reshape(9, 4, 16, 4, 8)                  # tap, ct, row, fk, j   (cin = fk * 8 + j)
            return np.ascontiguousarray(w.transpose(0, 1, 3, 2, 4)).astype(np.float16).reshape(-1)     # [tap][ct][fk*16+row][8]

        def pack_proj():
            w = proj['w'][0].reshape(4, 16, 4, 8)                     # (1 tap) ct, row, fk, j
            return np.ascontiguousarray(w.transpose(0, 2, 1, 3)).astype(np.float16).reshape(-1)
        f32 = lambda v: self._blob(np.ascontiguousarray(v, np.float32))
        x, oc, op = pool['inp'], conv['out'][0], proj['out'][0]
        _lib.check(self.lib.rtm3d_op_conv32s2_fused(self.ctx, self.tids[x.tid], x.coff, self.tids[oc.tid], oc.coff, self.tids[op.tid], op.coff,
                                                    self._blob(self._packed(conv, 0, 'c32s2', 0, pack_conv)), f32(conv['bias'][0]),
                                                    self._blob(self._packed(proj, 0, 'c32p', 0, pack_proj)), f32(proj['bias'][0])),
                   'op_conv32s2_fused')

    @staticmethod
    def _reads(op, tid):
        if op['op'] == 'conv':
            return any(s.tid == tid for s in op['inp']) or any(r is not None and r.tid == tid for r in op['res'])
        if op['op'] in ('maxpool', 'headout'):
            return op['inp'].tid == tid
        if op['op'] == 'patch_mask':
            return op['t'].tid == tid
        if op['op'] == 'softmax':
            return op['z_in'].tid == tid or any(u.tid == tid for u in op['us'])
        return False

    def _op_stem_fused(self, a, b, c=None):
        wb = self._packed(a, 0, 'smallc0', 0, lambda: pack_smallc_weights(a['w'][0], rows=False))
        wl = self._packed(b, 0, 'smallc0', 0, lambda: pack_smallc_weights(b['w'][0], rows=False))
        f32 = lambda v: self._blob(np.ascontiguousarray(v, np.float32))
        last = c if c is not None else b
        out = last['out'][0]
        w1, b1 = -1, -1
        if c is not None:
            w1 = self._blob(self._packed(c, 0, 'smallc0', 0, lambda: pack_smallc_weights(c['w'][0], rows=False)))
            b1 = f32(c['bias'][0])
        _lib.check(self.lib.rtm3d_op_stem_fused(self.ctx, self.tids[a['inp'][0].tid], self.tids[out.tid], out.coff,
                                                self._blob(wb), f32(a['bias'][0]), self._blob(wl), f32(b['bias'][0]), w1, b1),
                   'op_stem_fused')

    def _softmax_stat_producers(self):
        """{conv op index: slot} for the convolutions whose epilogue can emit the spatial-softmax partials of a
        fusion operand (rtm3d_conv_desc.softmax_stat_slot): every operand of a fusion must be the whole 256-channel
        output of a conv that takes the halo-tile kernel (stride 1, taps within +-1 pixel, 8x32 tiles cover the
        iteration domain, kernel variant 2), otherwise the fusion keeps its own reduction pass."""
        P = self.plan
        slots = {}
        for si, sop in enumerate(P.ops):
            if sop['op'] != 'softmax':
                continue
            prods = []
            for u in sop['us']:
                cand = [k for k in range(si) if P.ops[k]['op'] == 'conv'
                        and any(o is not None and o.tid == u.tid for o in P.ops[k]['out'])]
                if not cand or u.coff != 0 or u.C != 256 or P.tensors[u.tid]['C'] != 256:
                    prods = None
                    break
                prods.append(cand[-1])
            if prods is None or len(set(prods)) != len(prods) or len(prods) > 3:
                continue
            ok = True
            for k in prods:
                op = P.ops[k]
                G = op['groups']
                M = P.B * op['Hm'] * op['Wm']
                variant = op.get('variant')
                if variant is None:
                    variant = choose_variant(op['cin'], op['cout'], M, G, op['out_nchw'], len(op['taps'][0]), op['in_stride'])
                ok = ok and variant == 2 and op['cout'] == 256 and G <= 4 and op['in_stride'] == 1 and not op['out_nchw']
                ok = ok and op['Hm'] % 8 == 0 and op['Wm'] % 32 == 0 and len(op['taps'][0]) in (4, 9)
                ok = ok and all(r is None for r in op['res']) and all(o.coff == 0 and o.C == 256 for o in op['out'])
                ok = ok and all(abs(dy) <= 1 and abs(dx) <= 1 for tp in op['taps'] for dy, dx in tp)
                ok = ok and P.tensors[op['inp'][0].tid]['pad'] >= 1 and op['cin'] % 64 == 0 and op['cin'] * len(op['taps'][0]) // 64 >= 4
            if ok:
                for i, k in enumerate(prods):
                    slots[k] = i
        return slots

    def _packed(self, op, g, kind, bn, make):
        """Packed weights of (layer, group, kernel variant): from the model's WeightCache when there is one."""
        if self.cache is None or not op.get('name'):
            return make()
        return self.cache.get('pack:%s|%d|%s|%d' % (op['name'], g, kind, bn), make)

    def _blob(self, arr):
        arr = np.ascontiguousarray(arr)
        bid = ctypes.c_int()
        _lib.check(self.lib.rtm3d_blob_create(self.ctx, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, ctypes.byref(bid)), 'blob_create')
        return bid.value

    def _op_conv(self, op):
        # what this recorded conv stores as fp16 (the REALIZED weights: composed neck taps A_up W, summed fold biases ...):
        # Model.check_range reports them next to the un-rewritten plan's rows (verify.range_report(realized=...))
        self.weight_ranges.append({'op': op['name'], 'max_abs_w': float(np.abs(op['w']).max()) if np.size(op['w']) else 0.0,
                                   'max_abs_bias': float(np.abs(op['bias']).max()) if np.size(op['bias']) else 0.0})
        d = _lib.ConvDesc()
        G = op['groups']
        d.in_tensor = self.tids[op['inp'][0].tid]
        d.out_tensor = self.tids[op['out'][0].tid] if op['out'][0] is not None else -1
        d.res_tensor = self.tids[op['res'][0].tid] if op['res'][0] is not None else -1
        d.Hm, d.Wm, d.in_stride, d.out_scale = op['Hm'], op['Wm'], op['in_stride'], op['out_scale']
        d.cin, d.cout, d.groups, d.ntaps = op['cin'], op['cout'], G, len(op['taps'][0])
        for g in range(G):
            assert op['inp'][g].tid == op['inp'][0].tid
            d.in_coff[g] = op['inp'][g].coff
            d.out_coff[g] = op['out'][g].coff if op['out'][g] is not None else 0
            d.res_coff[g] = op['res'][g].coff if op['res'][g] is not None else 0
            d.out_oy[g], d.out_ox[g] = op['out_off'][g]
            for t, (dy, dx) in enumerate(op['taps'][g]):
                d.tap_dy[g][t], d.tap_dx[g][t] = dy, dx
                d.tap_dc[g][t] = op['tap_dc'][g][t] if 'tap_dc' in op else 0
        d.relu = 1 if op['relu'] else 0
        d.softmax_stat_slot = self._stat_slots.get(self._k, -1)
        d.out_nchw_f32 = op['out_nchw']
        d.out_H, d.out_W = op['out_hw']
        d.s2d_tensor, d.s2d_coff, d.in_s2d = 0, 0, 0              # s2d_tensor: tensor id + 1, 0 = none
        M = self.plan.B * op['Hm'] * op['Wm']
        variant = op.get('variant')
        s2d = getattr(self, '_s2d_for', {}).get(self._k)
        if s2d is not None and 'tap_dc' in op:
            #  the producer of a space-to-depth copy was itself rewritten with per-tap channel offsets: no kernel with the
            # second store takes tap_dc, so the copy would silently stay zero and the neck fold behind it would read zeros
            raise RuntimeError('plan: %s must write a space-to-depth copy of its output but was rewritten with tap_dc (project fold); '
                               'the two level rewrites exclude each other' % op['name'])
        if self._k in getattr(self, '_s2d_only_producers', ()) and s2d is None:
            raise RuntimeError('plan: %s is marked as writing only a space-to-depth copy but no copy was planned for it' % op['name'])
        if s2d is not None:
            # this conv's output feeds a neck up-fold: second copy in space-to-depth layout, 128-pixel kernel (the one whose epilogue has it)
            d.s2d_tensor, d.s2d_coff = self.tids[s2d[0].tid] + 1, s2d[1]
            variant = 5 if conv64_eligible(op) else 0
            if self._k in getattr(self, '_s2d_only_producers', ()):
                d.out_tensor = -1                 # every reader takes the copy: the ordinary output is not written
        if variant is None:
            variant = (5 if conv64_eligible(op) else 7 if conv64s2_eligible(op) and self.plan.B * (op['Hm'] // 4) * (op['Wm'] // 32) >= 64
                       else 6 if USE_CONV128 and conv128_eligible(op, self.plan.B)
                       else choose_variant(op['cin'], op['cout'], M, G, op['out_nchw'], len(op['taps'][0]), op['in_stride']))
        if self._k in getattr(self, '_in_s2d_for', {}) and variant != 7:
            #  this conv's input exists ONLY as its space-to-depth copy, which kernel 7 alone can read
            raise RuntimeError('plan: the input of %s exists only as a space-to-depth copy but the conv went to kernel variant %r' % (op['name'], variant))
        if variant == 7:
            assert conv64s2_eligible(op), op['name']
            d.kernel, d.bn_tile = 7, 128
            src = getattr(self, '_in_s2d_for', {}).get(self._k)
            if src is not None:                   # the producer wrote only the space-to-depth copy of this conv's input
                d.in_tensor, d.in_coff[0], d.in_s2d = self.tids[src[0].tid], src[1], 1
            d.w_blob = self._blob(self._packed(op, 0, 'c64s2', 0, lambda: pack_conv64s2_weights(op['w'][0])))
            d.bias_blob = self._blob(np.ascontiguousarray(op['bias'][0], np.float32))
        elif variant == 6:
            d.kernel, d.bn_tile = 6, 128
            d.w_blob = self._blob(self._packed(op, 0, 'mfma', 128, lambda: pack_mfma_weights(op['w'][0], 128)[0]))
            d.bias_blob = self._blob(np.ascontiguousarray(op['bias'][0], np.float32))
        elif variant == 5:
            assert conv64_eligible(op), op['name']
            d.kernel, d.bn_tile = 5, 64
            d.w_blob = self._blob(self._packed(op, 0, 'c64', 0, lambda: pack_conv64_weights(op['w'][0])))
            d.bias_blob = self._blob(np.ascontiguousarray(op['bias'][0], np.float32))
        elif variant == 2:
            packed = [self._packed(op, g, 'mfma', 256, lambda g=g: pack_mfma_weights(op['w'][g], 256)[0]) for g in range(G)]
            d.kernel, d.bn_tile = 2, 256
            d.w_blob, d.bias_blob = self._blob(np.concatenate(packed)), self._blob(np.ascontiguousarray(op['bias'], np.float32).reshape(-1))
        elif variant == 3:
            assert G == 1
            d.kernel, d.bn_tile = 3, 0
            rows = op['cin'] == 16 and op['cout'] == 16 and op['in_stride'] == 1 and op['out_scale'] == 1 and len(op['taps'][0]) == 9
            d.w_blob = self._blob(self._packed(op, 0, 'smallc%d' % rows, 0, lambda: pack_smallc_weights(op['w'][0], rows=rows)))
            d.bias_blob = self._blob(op['bias'][0])
        elif op['cin'] % 64 == 0:
            bn = op.get('bn_tile') or BN_TILE_OVERRIDE.get(op['name']) or choose_bn_tile(op['cout'], M)
            packed, biases = [], []
            for g in range(G):
                pw = self._packed(op, g, 'mfma', bn, lambda g=g: pack_mfma_weights(op['w'][g], bn)[0])
                cout_pad = (op['cout'] + bn - 1) // bn * bn
                packed.append(pw)
                bb = np.zeros(cout_pad, np.float32)
                bb[:op['cout']] = op['bias'][g]
                biases.append(bb)
            d.kernel, d.bn_tile = 0, bn
            d.w_blob, d.bias_blob = self._blob(np.concatenate(packed)), self._blob(np.concatenate(biases))
        else:
            raise NotImplementedError('no HIP kernel for conv %s (cin=%d)' % (op['name'], op['cin']))
        _lib.check(self.lib.rtm3d_op_conv(self.ctx, ctypes.byref(d)), 'op_conv ' + op['name'])

    def _op_headout(self, op):
        w, b = pack_headout_weights(op['w'], op['bias'])
        co = (ctypes.c_int * 4)(*([x.shape[0] for x in op['w']] + [0] * (4 - len(op['w']))))
        _lib.check(self.lib.rtm3d_op_headout(self.ctx, self.tids[op['inp'].tid], self._blob(w), self._blob(b), len(op['w']), co), 'op_headout')

    def _op_input4(self, op):
        _lib.check(self.lib.rtm3d_op_input_nhwc4(self.ctx, self.tids[op['out'].tid]), 'op_input_nhwc4')

    def _op_maxpool(self, op):
        _lib.check(self.lib.rtm3d_op_maxpool(self.ctx, self.tids[op['inp'].tid], op['inp'].coff, self.tids[op['out'].tid],
                                             op['out'].coff, op['inp'].C, op['k'], op['stride'], op['pad']), 'op_maxpool ' + op['name'])

    def _op_patch_mask(self, op):
        if getattr(self, 'yx_blob', None) is None:
            self.yx_blob = self._blob(np.full((self.plan.B, 2), -1, np.int32))
        Hm, Wm = self.plan.map_hw
        _lib.check(self.lib.rtm3d_op_patch_mask(self.ctx, self.tids[op['t'].tid], self.yx_blob, Hm, Wm, op['origin']), 'op_patch_mask')

    def tensor_info(self, s):
        """(device address of padded element [0][0][0][0], B, H, W, C, border) of the tensor a Slice lives in."""
        base = ctypes.c_void_p()
        v = [ctypes.c_int() for _ in range(5)]
        _lib.check(self.lib.rtm3d_tensor_info(self.ctx, self.tids[s.tid], ctypes.byref(base), *[ctypes.byref(x) for x in v]), 'tensor_info')
        return (base.value,) + tuple(x.value for x in v)

    def blob_address(self, bid):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(self.lib.rtm3d_blob_address(self.ctx, bid, ctypes.byref(p), ctypes.byref(n)), 'blob_address')
        return p.value

    def blob_bytes(self, bid):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(self.lib.rtm3d_blob_address(self.ctx, bid, ctypes.byref(p), ctypes.byref(n)), 'blob_address')
        return int(n.value)

    def _op_softmax(self, op):
        us = (ctypes.c_int * len(op['us']))(*[self.tids[u.tid] for u in op['us']])
        _lib.check(self.lib.rtm3d_op_softmax_fuse(self.ctx, self.tids[op['z_in'].tid], self.tids[op['z_out'].tid], len(op['us']), us),
                   'op_softmax_fuse')

    # ---- execution
    def forward(self, stream, d_in, d_out4):
        outs = (ctypes.c_void_p * 4)(*d_out4)
        _lib.check(self.lib.rtm3d_forward(self.ctx, ctypes.c_void_p(stream), ctypes.c_void_p(d_in), outs), 'forward')

    def forward_timed(self, stream, d_in, d_out4):
        outs = (ctypes.c_void_p * 4)(*d_out4)
        n = ctypes.c_int()
        _lib.check(self.lib.rtm3d_forward_timed(self.ctx, ctypes.c_void_p(stream), ctypes.c_void_p(d_in), outs, None, 0, ctypes.byref(n)), 'forward_timed')
        ms = (ctypes.c_float * n.value)()
        _lib.check(self.lib.rtm3d_forward_timed(self.ctx, ctypes.c_void_p(stream), ctypes.c_void_p(d_in), outs, ms, n.value, ctypes.byref(n)), 'forward_timed')
        info = []
        for i in range(n.value):
            fl, by, nm = ctypes.c_double(), ctypes.c_double(), ctypes.c_char_p()
            _lib.check(self.lib.rtm3d_op_info(self.ctx, i, ctypes.byref(fl), ctypes.byref(by), ctypes.byref(nm)), 'op_info')
            info.append({'kernel': nm.value.decode(), 'name': self.op_names[i], 'ms': float(ms[i]), 'flops': fl.value, 'bytes': by.value})
        return info

    def forward_marks(self, stream, d_in, d_out4, mark_ops):
        """Wall ms of the stages that start at the recorded ops `mark_ops` (ascending) in ONE real eager replay
        (rtm3d_forward_marks): [mark i -> mark i + 1 ..., last mark -> end]."""
        outs = (ctypes.c_void_p * 4)(*d_out4)
        m = (ctypes.c_int * len(mark_ops))(*mark_ops)
        ms = (ctypes.c_float * len(mark_ops))()
        _lib.check(self.lib.rtm3d_forward_marks(self.ctx, ctypes.c_void_p(stream), ctypes.c_void_p(d_in), outs, len(mark_ops), m, ms), 'forward_marks')
        return [float(v) for v in ms]

    def kernel_names(self):
        """Kernel name of every recorded runtime op (rtm3d_op_info), in launch order."""
        names = []
        for i in range(len(self.op_names)):
            fl, by, nm = ctypes.c_double(), ctypes.c_double(), ctypes.c_char_p()
            _lib.check(self.lib.rtm3d_op_info(self.ctx, i, ctypes.byref(fl), ctypes.byref(by), ctypes.byref(nm)), 'op_info')
            names.append(nm.value.decode())
        return names

    def input_tensor(self):
        """(device address, border) of the fp16 NHWC4 tensor the stem reads (rtm3d_input_tensor)."""
        base, B, H, W, P = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.rtm3d_input_tensor(self.ctx, ctypes.byref(base), ctypes.byref(B), ctypes.byref(H), ctypes.byref(W),
                                               ctypes.byref(P)), 'input_tensor')
        assert (B.value, H.value, W.value) == (self.plan.B, self.plan.H, self.plan.W)
        return base.value, P.value

    def set_graph(self, enable):
        """Replay the plan as one hipGraph launch (bit-identical results; see rtm3d_ctx_set_graph)."""
        _lib.check(self.lib.rtm3d_ctx_set_graph(self.ctx, 1 if enable else 0), 'ctx_set_graph')

    def graph_stats(self):
        """(captures, cache hits, graph mode still enabled) of this context (rtm3d_ctx_graph_stats)."""
        c, h, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.rtm3d_ctx_graph_stats(self.ctx, ctypes.byref(c), ctypes.byref(h), ctypes.byref(e)), 'ctx_graph_stats')
        return c.value, h.value, bool(e.value)

    def probe_set(self, op_index):
        _lib.check(self.lib.rtm3d_probe_set(self.ctx, int(op_index)), 'probe_set')

    def probe_read(self):
        ms, n = ctypes.c_double(), ctypes.c_int()
        _lib.check(self.lib.rtm3d_probe_read(self.ctx, ctypes.byref(ms), ctypes.byref(n)), 'probe_read')
        return ms.value, n.value

    def download(self, s):
        """Debug: fp32 NCHW copy of a Slice."""
        t = self.plan.tensors[s.tid]
        src = getattr(self, '_s2d_only', {}).get((s.tid, s.coff, s.C))
        if src is not None:
            # only the space-to-depth copy of this map exists on the device: gather its four phase slices back
            out = np.empty((self.plan.B, s.C, t['H'], t['W']), np.float32)
            ph = np.empty((self.plan.B, s.C, t['H'] // 2, t['W'] // 2), np.float32)
            for py in range(2):
                for px in range(2):
                    _lib.check(self.lib.rtm3d_tensor_download(self.ctx, self.tids[src[0].tid], src[1] + (py * 2 + px) * s.C, s.C,
                                                              ph.ctypes.data_as(ctypes.c_void_p)), 'tensor_download')
                    out[:, :, py::2, px::2] = ph
            return out
        out = np.empty((self.plan.B, s.C, t['H'], t['W']), np.float32)
        _lib.check(self.lib.rtm3d_tensor_download(self.ctx, self.tids[s.tid], s.coff, s.C, out.ctypes.data_as(ctypes.c_void_p)), 'tensor_download')
        return out

    def close(self):
        peak = getattr(self, 'peak', None)
        if peak is not None:
            peak.close()
            self.peak = None
        if self.ctx:
            self.lib.rtm3d_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
