"""CPU restatement (PyTorch-CPU fp32, functional) of the reference forward pass.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
file:line it follows (paths relative to /root/reference).  The restatement is driven by a
plain ``state_dict`` that uses the reference's own key names, so the very same tensors can be
loaded into the reference ``Model`` (that is how tests/golden/make_golden.py pins it).

Pinned by: tests/golden/*.npz  (tests/test_oracle_golden.py).
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-4  # utils/torch_utils.py:79-81 (initialize_weights overwrites every BatchNorm2d.eps)

DLA34_LEVELS = [1, 1, 1, 2, 2, 1]          # models/nets/dla.py:13-18
DLA34_CHANNELS = [16, 32, 64, 128, 256, 512]
RESNET_SPEC = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}   # models/nets/resnet.py:226-230 (BasicBlock variants)


def _bn(x, sd, p):
    # nn.BatchNorm2d in eval mode, eps from initialize_weights (utils/torch_utils.py:79-81)
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _conv(x, sd, p, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding, dilation)


# ----------------------------------------------------------------------------- DLA-34
def _dla_basic_block(x, sd, p, stride, residual=None):
    # models/nets/dla.py:86-100  BasicBlock.forward
    if residual is None:
        residual = x
    out = F.relu(_bn(_conv(x, sd, p + '.conv1', stride, 1), sd, p + '.norm1'))
    out = _bn(_conv(out, sd, p + '.conv2', 1, 1), sd, p + '.norm2')
    out = out + residual
    return F.relu(out)


def _dla_root(xs, sd, p):
    # models/nets/dla.py:233-241  Root.forward (kernel 1, residual=False)
    x = _conv(torch.cat(xs, 1), sd, p + '.conv')
    return F.relu(_bn(x, sd, p + '.norm'))


def _dla_tree(x, sd, p, level, cin, cout, stride, level_root, children=None):
    # models/nets/dla.py:186-210  Tree.forward ; ctor :103-184
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x          # :190-193
    if cin != cout:                                                         # :195-198, :175-184
        residual = _bn(_conv(bottom, sd, p + '.project.0'), sd, p + '.project.1')
    else:
        residual = bottom
    if level_root:                                                          # :200-201
        children.append(bottom)
    if level == 1:
        x1 = _dla_basic_block(x, sd, p + '.tree1', stride, residual)       # :202
        x2 = _dla_basic_block(x1, sd, p + '.tree2', 1)                      # :205
        return _dla_root([x2, x1] + children, sd, p + '.root')             # :206
    # nested tree: the `residual` argument handed to a Tree is recomputed inside it (:195-198)
    x1 = _dla_tree(x, sd, p + '.tree1', level - 1, cin, cout, stride, False)
    children.append(x1)                                                     # :208
    return _dla_tree(x1, sd, p + '.tree2', level - 1, cout, cout, 1, False, children)  # :209


def dla34_forward(x, sd, prefix='backbone'):
    """models/nets/dla.py:322-332 DLABase.forward -> [level2, level3, level4, level5]."""
    p = prefix
    ch = DLA34_CHANNELS
    x = F.relu(_bn(_conv(x, sd, p + '.base_layer.0', 1, 3), sd, p + '.base_layer.1'))   # :259-268
    x = F.relu(_bn(_conv(x, sd, p + '.level0.0', 1, 1), sd, p + '.level0.1'))           # :270-273
    x = F.relu(_bn(_conv(x, sd, p + '.level1.0', 2, 1), sd, p + '.level1.1'))           # :275-279
    ys = []
    for i in range(2, 6):                                                               # :281-315
        x = _dla_tree(x, sd, '%s.level%d' % (p, i), DLA34_LEVELS[i], ch[i - 1], ch[i], 2,
                      level_root=(i > 2))
        ys.append(x)
    return ys


# ----------------------------------------------------------------------------- ResNet
def _res_basic_block(x, sd, p, stride, has_down):
    # models/nets/resnet.py:55-72
    out = F.relu(_bn(_conv(x, sd, p + '.conv1', stride, 1), sd, p + '.bn1'))
    out = _bn(_conv(out, sd, p + '.conv2', 1, 1), sd, p + '.bn2')
    residual = x
    if has_down:                                                                        # :145-151
        residual = _bn(_conv(x, sd, p + '.downsample.0', stride, 0), sd, p + '.downsample.1')
    return F.relu(out + residual)


def resnet_forward(x, sd, num_layers=18, prefix='backbone'):
    """models/nets/resnet.py:200-211 PoseResNet.forward -> [layer1..layer4]."""
    p = prefix
    x = F.relu(_bn(_conv(x, sd, p + '.conv1', 2, 3), sd, p + '.bn1'))                   # :124-126
    x = F.max_pool2d(x, 3, 2, 1)                                                        # :128
    ys = []
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip([64, 128, 256, 512], RESNET_SPEC[int(num_layers)])):
        stride = 1 if li == 0 else 2
        for b in range(blocks):
            s = stride if b == 0 else 1
            down = (b == 0) and (s != 1 or inplanes != planes)
            x = _res_basic_block(x, sd, '%s.layer%d.%d' % (p, li + 1, b), s, down)
            inplanes = planes
        ys.append(x)
    return ys


# ----------------------------------------------------------------------------- neck
def _up(x, sd, p):
    # models/nets/module.py:7-15 ConvTranspose2d(c, c, 4, stride=2, padding=1, bias=False)
    return F.conv_transpose2d(x, sd[p + '.conv_tran.weight'], None, 2, 1)


def kfpn_fusion_forward(xs, sd, levels=(2, 3, 4, 5), prefix='kfpn_fusion'):
    """models/nets/keypoint_fpn_fusion.py:35-69 (_fpn + forward)."""
    p = prefix
    x = list(xs)
    n = len(levels)
    for i in range(n - 1, 0, -1):                                                       # :37-43
        L = levels[i]
        x[i] = _conv(x[i], sd, '%s.kfpn_head%d' % (p, L))
        up = _up(x[i], sd, '%s.kfpn_up%d' % (p, L))
        x[i - 1] = _conv(torch.cat([up, x[i - 1]], 1), sd, '%s.kfpn_proj%d' % (p, L))
    x[0] = _conv(x[0], sd, '%s.kfpn_head%d' % (p, levels[0]))                           # :44-45
    z = x[0]
    for i in range(n - 1, 0, -1):                                                       # :61-68
        u = x[i]
        for j in range(levels[i] - levels[0]):
            u = _up(u, sd, '%s.fusion_up%d.%d' % (p, levels[i], j))
        bs, c, h, w = u.shape
        z = z + u * torch.softmax(u.view(bs, c, -1), dim=-1).view(bs, c, h, w)          # :66-68
    return z


# ----------------------------------------------------------------------------- heads
HEADS = [('main_kf_header', 'main_kf_head'), ('offset_fr_main_header', 'offset_fr_main_head'),
         ('main_offset_header', 'main_offset_head'), ('vertex_offset_header', 'vertex_offset_head')]


def header_forward(z, sd, prefix='detect_header'):
    """models/nets/header.py:40-46; branch layout :13-37 + utils/torch_utils.py:179-204."""
    outs = []
    for seq, last in HEADS:
        p = '%s.%s' % (prefix, seq)
        h = F.relu(_bn(_conv(z, sd, p + '.0', 1, 6, 6), sd, p + '.1'))    # conv3x3 d6 p6 + bias, BN, ReLU
        k = 1                                                             # HEADER_NUM_CONV - 1 further levels (header.py:12: dilation [6] + [1] * (n - 1);
        while '%s.%d.weight' % (p, 3 * k) in sd:                          #  make_conv_level numbers them 3k, 3k + 1, 3k + 2 in the Sequential)
            h = F.relu(_bn(_conv(h, sd, '%s.%d' % (p, 3 * k), 1, 1, 1), sd, '%s.%d' % (p, 3 * k + 1)))   # conv3x3 d1 p1 + bias, BN, ReLU
            k += 1
        outs.append(_conv(h, sd, '%s.%s' % (p, last), 1, 1, 1))           # conv3x3 p1 -> C_out + bias
    return tuple(outs)


# ----------------------------------------------------------------------------- 2D decode
def nms_hm(heat_map, kernel=3):
    # utils/model_utils.py:17-26
    pad = (kernel - 1) // 2
    hmax = F.max_pool2d(heat_map, (kernel, kernel), stride=1, padding=pad)
    return heat_map * (hmax == heat_map).float()


def obtain_main_proj2d(main_kf_logits, confidence, topk):
    # models/model.py:77-98 (operates on a clone; the reference mutates its argument)
    hm = torch.sigmoid(main_kf_logits)
    hm = nms_hm(hm.unsqueeze(0), 3).squeeze(0)
    K, H, W = hm.shape
    scores, indices = torch.topk(hm.reshape(-1), topk, dim=0)
    keep = scores > confidence
    scores, indices = scores[keep], indices[keep]
    cls = indices // (H * W)
    xy = indices % (H * W)
    return cls, scores, [(xy % W).to(torch.float32), (xy // W).to(torch.float32)]


def inference(pred_logits, score_thresh=0.4, topk=100, down_sample=4.0):
    """models/model.py:29-75 Model.inference (per-image Python loop)."""
    main_kf, offset_fr_main, main_offset, _vertex_offset = pred_logits
    bs = main_kf.shape[0]
    clses, m_scores, m_projs, v_projs_regress, bboxes_2d = ([None] * bs for _ in range(5))
    for i in range(bs):
        cls_i, sc_i, (x_i, y_i) = obtain_main_proj2d(main_kf[i], score_thresh, topk)
        if len(cls_i) == 0:                                                             # :43-44
            continue
        n = len(x_i)
        yl, xl = y_i.long(), x_i.long()
        off = offset_fr_main[i][:, yl, xl].view(-1, 2, n).permute(0, 2, 1).contiguous()  # :124-128 (8,N,2)
        sub = torch.sigmoid(main_offset[i][:, yl, xl])                                  # :48
        x_i = x_i + sub[0]                                                              # :49-50
        y_i = y_i + sub[1]
        mp = torch.cat([x_i.unsqueeze(-1), y_i.unsqueeze(-1)], dim=-1)                  # :62
        vr = off.permute(1, 0, 2).contiguous() + mp.view(-1, 1, 2)                      # :63
        clses[i] = cls_i
        m_scores[i] = sc_i
        m_projs[i] = down_sample * mp                                                   # :66
        v_projs_regress[i] = down_sample * vr                                           # :69
        bboxes_2d[i] = torch.cat([v_projs_regress[i].min(dim=1)[0],
                                  v_projs_regress[i].max(dim=1)[0]], dim=-1)            # :70-72
    return clses, m_scores, m_projs, v_projs_regress, bboxes_2d


# ----------------------------------------------------------------------------- whole model
def model_forward(x, sd, backbone='DLA-34', score_thresh=0.4, topk=100, down_sample=4.0,
                  return_stages=False):
    """models/model.py:20-27 Model.forward in eval mode."""
    with torch.no_grad():
        if 'DLA' in backbone:
            feats = dla34_forward(x, sd)
        else:
            feats = resnet_forward(x, sd, int(backbone.split('-')[-1]))
        z = kfpn_fusion_forward(feats, sd)
        logits = header_forward(z, sd)
        dets = inference([l.clone() for l in logits], score_thresh, topk, down_sample)
    if return_stages:
        return dets, logits, {'feats': feats, 'z': z}
    return dets, logits
