"""CPU restatement of the "smoke" head-table variant.

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: the smoke branch of the reference is not part of the
snapshot under /root/reference (only README.md:2-4 mentions it), so there is no reference source, test
or golden vector to check this against.  It restates the published SMOKE formulation (Liu, Wu, Toth:
"SMOKE: Single-Stage Monocular 3D Object Detection via Keypoint Estimation", 2020) on top of the
reference's own backbone / neck / head stack and main-key-point decode, and serves as a self-consistency
check of the HIP path for that head layout.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import rtm3d_ref

HEADS_SMOKE = [('main_kf_header', 'main_kf_head'), ('regression_header', 'regression_head')]
DEPTH_REF = (28.01, 16.32)


def model_forward(x, sd, backbone='DLA-34'):
    """backbone + neck of the reference, then the two 3-conv branches (same stack as header.py:13-37)."""
    with torch.no_grad():
        feats = rtm3d_ref.dla34_forward(x, sd) if 'DLA' in backbone else rtm3d_ref.resnet_forward(x, sd, int(backbone.split('-')[-1]))
        z = rtm3d_ref.kfpn_fusion_forward(feats, sd)
        outs = []
        for seq, last in HEADS_SMOKE:
            p = 'detect_header.' + seq
            h = F.relu(rtm3d_ref._bn(rtm3d_ref._conv(z, sd, p + '.0', 1, 6, 6), sd, p + '.1'))
            h = F.relu(rtm3d_ref._bn(rtm3d_ref._conv(h, sd, p + '.3', 1, 1, 1), sd, p + '.4'))
            outs.append(rtm3d_ref._conv(h, sd, '%s.%s' % (p, last), 1, 1, 1))
    return tuple(outs)


def decode(main_kf, reg, K, dim_ref, score_thresh=0.4, topk=100, down=4.0):
    """Per image: main key points as in models/model.py:77-98, then the closed-form SMOKE box.
    Returns list of None | dict(cls, score, xy, x8) with x8 = [sin ry, cos ry, l, h, w, X, Y, Z] (fp64)."""
    out = []
    K = np.asarray(K, np.float64).reshape(-1, 9)
    for i in range(main_kf.shape[0]):
        cls, sc, (xs, ys) = rtm3d_ref.obtain_main_proj2d(main_kf[i], score_thresh, topk)
        if len(cls) == 0:
            out.append(None)
            continue
        r = reg[i][:, ys.long(), xs.long()].numpy().astype(np.float64)        # (8, N)
        Kb = K[i if K.shape[0] > 1 else 0]
        z = DEPTH_REF[0] + DEPTH_REF[1] * r[0]
        u = down * (xs.numpy().astype(np.float64) + r[1]); v = down * (ys.numpy().astype(np.float64) + r[2])
        X = (u - Kb[2]) * z / Kb[0]; Y = (v - Kb[5]) * z / Kb[4]
        dr = np.asarray(dim_ref, np.float64)[cls.numpy()]
        h, w, l = dr[:, 0] * np.exp(r[3]), dr[:, 1] * np.exp(r[4]), dr[:, 2] * np.exp(r[5])
        alpha = np.arctan(r[6] / (r[7] + 1e-7)) + np.where(r[7] >= 0, -0.5 * np.pi, 0.5 * np.pi)
        ry = alpha + np.arctan2(X, z)
        ry = np.where(ry > np.pi, ry - 2 * np.pi, ry); ry = np.where(ry < -np.pi, ry + 2 * np.pi, ry)
        out.append({'cls': cls.numpy(), 'score': sc.numpy(), 'xy': np.stack([xs.numpy(), ys.numpy()], 1),
                    'x8': np.stack([np.sin(ry), np.cos(ry), l, h, w, X, Y, z], 1)})
    return out
