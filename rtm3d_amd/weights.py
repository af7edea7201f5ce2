"""State-dict layout of the reference model and seeded synthetic weights.

The key names and tensor shapes reproduce ``Model(cfg, backbone).state_dict()`` of the
reference (models/model.py:10-18, models/nets/dla.py:244-320, models/nets/resnet.py:116-158,
models/nets/keypoint_fpn_fusion.py:8-33, models/nets/header.py:6-37) so a reference
checkpoint's ``ckpt["model"]`` loads unchanged (utils/check_point.py:80-92).

There is no network access for the published checkpoints, so benchmarks and tests use
``synth_state_dict``: a deterministic (numpy PCG64) generator.  Two styles:

* ``"init"``    - what ``initialize_weights`` leaves behind (utils/torch_utils.py:71-83):
  Xavier-uniform convs, bilinear channel-0 deconvs, identity BatchNorm statistics.
* ``"trained"`` - "trained-like": variance-preserving conv gains, randomised BatchNorm
  affine/statistics, conv biases, and a negative heat-map bias so the key-point heat map is
  sparse (a handful of detections per image) and activations stay O(1) in fp16.
"""
import math
import re
from collections import OrderedDict

import numpy as np
import torch

DLA34_LEVELS = [1, 1, 1, 2, 2, 1]
DLA34_CHANNELS = [16, 32, 64, 128, 256, 512]
RESNET_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}
HEADS = [('main_kf_header', 'main_kf_head', 3), ('offset_fr_main_header', 'offset_fr_main_head', 16),
         ('main_offset_header', 'main_offset_head', 2), ('vertex_offset_header', 'vertex_offset_head', 2)]
# "smoke" head-table variant (the branch's source is NOT in the reference
# snapshot -> parity unpinned): rtm3d's main key-point heat map + ONE regression branch with the 8
# channels of the SMOKE paper [dz, dxc, dyc, dh, dw, dl, sin(alpha), cos(alpha)], same 3-conv stack.
HEADS_SMOKE = [('main_kf_header', 'main_kf_head', 3), ('regression_header', 'regression_head', 8)]


def head_table(variant='rtm3d', num_classes=3):
    """[(branch, last conv, channels)]; the heat-map branch has one channel per class of cfg.DATASET.OBJs
    (models/nets/header.py:11)."""
    if not 1 <= int(num_classes) <= 16:
        raise ValueError('the heat-map head supports 1..16 classes (conv_headout.hip), got %d' % num_classes)
    if variant in (None, 'rtm3d'):
        return [(HEADS[0][0], HEADS[0][1], int(num_classes))] + HEADS[1:]
    if variant == 'smoke':
        return [(HEADS_SMOKE[0][0], HEADS_SMOKE[0][1], int(num_classes))] + HEADS_SMOKE[1:]
    raise ValueError('unknown MODEL.HEAD_VARIANT %r' % (variant,))


def parse_backbone(name):
    """'DLA-34' -> ('dla', 34); 'RESNET-18' -> ('resnet', 18)  (models/model_factory.py:26-35)."""
    name = str(name)
    if 'DLA-34' in name:
        return 'dla', 34
    if 'RESNET' in name:
        n = int(name.split('-')[-1])
        if n not in RESNET_BLOCKS:
            # The reference itself cannot run these: PoseResNet._kfpn_spec (models/nets/resnet.py:129-137) ignores
            # Bottleneck.expansion, so the neck's 1x1 convs are built for 64..512 input channels and receive
            # 256..2048 (RuntimeError in kfpn_head on the first forward).  There is no behaviour to match.
            raise NotImplementedError('only the BasicBlock ResNets (18/34) exist as runnable models in the reference; '
                                      '%s raises in its neck there (kfpn_spec ignores Bottleneck.expansion)' % name)
        return 'resnet', n
    raise AssertionError('Undefined model backbone')


class _Spec(object):
    def __init__(self):
        self.items = []   # (key, shape, kind, meta)

    def conv(self, p, cout, cin, k, bias=False):
        self.items.append((p + '.weight', (cout, cin, k, k), 'conv', {'bias': bias}))
        if bias:
            self.items.append((p + '.bias', (cout,), 'conv_bias', {}))

    def deconv(self, p, c):
        self.items.append((p + '.weight', (c, c, 4, 4), 'deconv', {}))

    def bn(self, p, c):
        self.items.append((p + '.weight', (c,), 'bn_weight', {}))
        self.items.append((p + '.bias', (c,), 'bn_bias', {}))
        self.items.append((p + '.running_mean', (c,), 'bn_mean', {}))
        self.items.append((p + '.running_var', (c,), 'bn_var', {}))
        self.items.append((p + '.num_batches_tracked', (), 'bn_count', {}))


def _dla_tree_spec(s, p, level, cin, cout, stride, level_root, root_dim=0):
    # registration order follows Tree.__init__ (models/nets/dla.py:103-184): tree1, tree2, root, project
    if root_dim == 0:
        root_dim = 2 * cout
    if level_root:
        root_dim += cin
    if level == 1:
        for t, ci in (('tree1', cin), ('tree2', cout)):
            s.conv('%s.%s.conv1' % (p, t), cout, ci, 3)
            s.bn('%s.%s.norm1' % (p, t), cout)
            s.conv('%s.%s.conv2' % (p, t), cout, cout, 3)
            s.bn('%s.%s.norm2' % (p, t), cout)
        s.conv(p + '.root.conv', cout, root_dim, 1)
        s.bn(p + '.root.norm', cout)
    else:
        _dla_tree_spec(s, p + '.tree1', level - 1, cin, cout, stride, False, 0)
        _dla_tree_spec(s, p + '.tree2', level - 1, cout, cout, 1, False, root_dim + cout)
    if cin != cout:
        s.conv(p + '.project.0', cout, cin, 1)
        s.bn(p + '.project.1', cout)


def state_dict_spec(backbone, head_variant='rtm3d', num_classes=3, header_num_conv=2):
    """Ordered [(key, shape, kind, meta)] identical to the reference ``state_dict()`` order."""
    kind, depth = parse_backbone(backbone)
    s = _Spec()
    if kind == 'dla':
        ch = DLA34_CHANNELS
        s.conv('backbone.base_layer.0', ch[0], 3, 7)
        s.bn('backbone.base_layer.1', ch[0])
        s.conv('backbone.level0.0', ch[0], ch[0], 3)
        s.bn('backbone.level0.1', ch[0])
        s.conv('backbone.level1.0', ch[1], ch[0], 3)
        s.bn('backbone.level1.1', ch[1])
        for i in range(2, 6):
            _dla_tree_spec(s, 'backbone.level%d' % i, DLA34_LEVELS[i], ch[i - 1], ch[i], 2, i > 2)
        feat_ch = ch[2:]
    else:
        s.conv('backbone.conv1', 64, 3, 7)
        s.bn('backbone.bn1', 64)
        inpl = 64
        for li, (pl, nb) in enumerate(zip([64, 128, 256, 512], RESNET_BLOCKS[depth])):
            for b in range(nb):
                p = 'backbone.layer%d.%d' % (li + 1, b)
                stride = 2 if (li > 0 and b == 0) else 1
                s.conv(p + '.conv1', pl, inpl, 3)
                s.bn(p + '.bn1', pl)
                s.conv(p + '.conv2', pl, pl, 3)
                s.bn(p + '.bn2', pl)
                if b == 0 and (stride != 1 or inpl != pl):
                    s.conv(p + '.downsample.0', pl, inpl, 1)
                    s.bn(p + '.downsample.1', pl)
                inpl = pl
        feat_ch = [64, 128, 256, 512]
    oc = 256
    # models/nets/keypoint_fpn_fusion.py:18-33 (levels 5,4,3 then head2, then fusion_up5,4,3)
    for i in (3, 2, 1):
        L = i + 2
        s.conv('kfpn_fusion.kfpn_head%d' % L, oc, feat_ch[i], 1, bias=True)
        s.deconv('kfpn_fusion.kfpn_up%d.conv_tran' % L, oc)
        s.conv('kfpn_fusion.kfpn_proj%d' % L, feat_ch[i - 1], feat_ch[i - 1] + oc, 1, bias=True)
    s.conv('kfpn_fusion.kfpn_head2', oc, feat_ch[0], 1, bias=True)
    for i in (3, 2, 1):
        for j in range(i):
            s.deconv('kfpn_fusion.fusion_up%d.%d.conv_tran' % (i + 2, j), oc)
    # models/nets/header.py:13-37
    for seq, last, cout in head_table(head_variant, num_classes):
        p = 'detect_header.' + seq
        # make_conv_level (utils/torch_utils.py:179-204): HEADER_NUM_CONV x (conv, BN, ReLU) = Sequential indices 3k, 3k + 1, (3k + 2)
        for k in range(int(header_num_conv)):
            s.conv('%s.%d' % (p, 3 * k), oc, oc, 3, bias=True)
            s.bn('%s.%d' % (p, 3 * k + 1), oc)
        s.conv('%s.%s' % (p, last), cout, oc, 3, bias=True)
    return s.items


def _bilinear_kernel(k=4):
    # utils/torch_utils.py:58-68 (_fill_up_weights)
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2. * f)
    w = np.zeros((k, k), np.float32)
    for i in range(k):
        for j in range(k):
            w[i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    return w


# Gains of the "trained" style relative to He-uniform, tuned (with the CPU oracle) so that
# feature maps, the fused map z and the logits stay O(1) through ~40 layers for both backbones.
_TRAINED_GAINS = {
    'dla': {'conv': 0.85, 'conv2': 0.5, 'neck': 0.8 * math.sqrt(0.5), 'deconv': 0.7, 'h0': 4.0, 'h3': 1.0, 'head': 1.0},
    'resnet': {'conv': 0.75, 'conv2': 0.45, 'neck': 0.7 * math.sqrt(0.5), 'deconv': 0.6, 'h0': 2.0, 'h3': 1.0, 'head': 1.0},
}


def _trained_gain(bkind, key):
    g = _TRAINED_GAINS[bkind]
    if '.kfpn_' in key:
        return g['neck']                       # linear 1x1 convs (no ReLU after)
    if key.startswith('detect_header'):
        if key.endswith('.0.weight'):
            return g['h0']
        if re.search(r'\.(3|6|9|12)\.weight$', key):       # the dilation-1 convs of the branch (HEADER_NUM_CONV - 1 of them)
            return g['h3']
        return g['head']
    if key.endswith('conv2.weight'):
        return g['conv2']                      # residual branch, added to the skip before ReLU
    return g['conv']


def synth_state_dict(backbone, seed=0, style='trained', heat_bias=-6.0, head_variant='rtm3d', heat_gain=1.0, num_classes=3, header_num_conv=2):
    """Deterministic synthetic weights (fp32 CPU tensors) under the reference key names.
    ``heat_gain`` scales the last heat-map conv ("trained" style only): > 1 spreads the peak scores over a wider
    range, as a trained detector's are, instead of the narrow band random features give."""
    assert style in ('init', 'trained')
    rng = np.random.Generator(np.random.PCG64(seed))
    bkind = parse_backbone(backbone)[0]
    sd = OrderedDict()
    bil = _bilinear_kernel(4)
    for key, shape, kind, meta in state_dict_spec(backbone, head_variant, num_classes, header_num_conv):
        if kind == 'conv':
            cout, cin, k, _ = shape
            fan_in, fan_out = cin * k * k, cout * k * k
            if style == 'init':
                a = math.sqrt(6.0 / (fan_in + fan_out))          # xavier_uniform_ (torch_utils.py:75)
            else:
                a = math.sqrt(6.0 / fan_in) * _trained_gain(bkind, key)
            v = rng.uniform(-a, a, size=shape).astype(np.float32)
            if style == 'trained' and heat_gain != 1.0 and key.endswith('main_kf_head.weight'):
                v = v * np.float32(heat_gain)
        elif kind == 'conv_bias':
            if style == 'init':
                bound = 1.0 / math.sqrt(256 * 9)
                v = rng.uniform(-bound, bound, size=shape).astype(np.float32)
            else:
                v = (0.05 * rng.standard_normal(shape)).astype(np.float32)
                if key.endswith('main_kf_head.bias'):
                    v = v + np.float32(heat_bias)
        elif kind == 'deconv':
            c = shape[0]
            if style == 'init':
                a = 1.0 / math.sqrt(shape[1] * 16)               # ConvTranspose2d default init bound
            else:
                a = math.sqrt(3.0 / (c * 4)) * _TRAINED_GAINS[bkind]['deconv']   # 4 taps x c inputs per output
            v = rng.uniform(-a, a, size=shape).astype(np.float32)
            v[:, 0, :, :] = bil                                  # _fill_up_weights touches w[c,0] only
        elif kind == 'bn_weight':
            v = np.ones(shape, np.float32) if style == 'init' else rng.uniform(0.6, 1.4, shape).astype(np.float32)
        elif kind == 'bn_bias':
            v = np.zeros(shape, np.float32) if style == 'init' else (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == 'bn_mean':
            v = np.zeros(shape, np.float32) if style == 'init' else (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif kind == 'bn_var':
            v = np.ones(shape, np.float32) if style == 'init' else rng.uniform(0.7, 1.3, shape).astype(np.float32)
        elif kind == 'bn_count':
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v))
    return sd


def synth_images(batch, height=384, width=1280, seed=1234, first=0):
    """Synthetic normalised images: image b is ``randn`` from PCG64(seed + first + b) so that
    shards are reproducible independently of the rank layout."""
    out = np.empty((batch, 3, height, width), np.float32)
    for b in range(batch):
        rng = np.random.Generator(np.random.PCG64(seed + first + b))
        out[b] = rng.standard_normal((3, height, width), dtype=np.float32)
    return torch.from_numpy(out)


def synth_intrinsics(pad_h=0.0):
    """KITTI P2 scaled by 1280/1242 (fx=fy=743.6, cx=628.2, cy=178.1+pad), row-major 3x3 as (9,)."""
    s = 1280.0 / 1242.0
    return np.array([721.5377 * s, 0.0, 609.5593 * s, 0.0, 721.5377 * s, 172.854 * s + pad_h, 0.0, 0.0, 1.0],
                    dtype=np.float64)
