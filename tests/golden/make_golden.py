#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Run only in the build container, where /root/reference exists (it does not travel):

    cd /tmp && python -B /root/repo/tests/golden/make_golden.py

This script contains no reference source.  It imports the reference's modules from
/root/reference (with `sys.modules` stubs for the absent `torchvision`, whose only use - the
DeformConv2d class - is dead code), loads seeded synthetic weights made by
``rtm3d_amd.weights.synth_state_dict`` into the reference ``Model``, runs it on synthetic inputs
and stores inputs + outputs as small .npz files.  tests/test_oracle_golden.py then checks the CPU
oracle (oracle/) against these files, and the `-m gpu` tests check the HIP path against them.
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'

tv = types.ModuleType('torchvision')
tvo = types.ModuleType('torchvision.ops')
tvm = types.ModuleType('torchvision.models')
tvo.DeformConv2d = type('DeformConv2d', (), {})
tv.ops, tv.models = tvo, tvm
sys.modules.update({'torchvision': tv, 'torchvision.ops': tvo, 'torchvision.models': tvm})
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402
from scipy.optimize import minimize  # noqa: E402

from models.nets import dla, resnet  # noqa: E402  (reference)
from models.model import Model  # noqa: E402       (reference)
from utils import model_utils as ref_mu  # noqa: E402 (reference)

from rtm3d_amd import weights  # noqa: E402

torch.set_num_threads(8)
DIM_REF = [[1.52607842, 1.62858147, 3.88396124], [1.76067766, 0.6602296, 0.84220464],
           [1.73712792, 0.59677122, 1.76338868]]          # models/configs/rtm3d_dla34_kitti.yaml:18-27
REF_LOC = [0, -0.5, 20]                                    # detect.py:74
KFNS = {'DLA-34': ['level2', 'level3', 'level4', 'level5'], 'RESNET-18': ['layer1', 'layer2', 'layer3', 'layer4'],
        'RESNET-34': ['layer1', 'layer2', 'layer3', 'layer4']}
# The 2D-decode vectors are bit-exact pins of ATen's CPU sigmoid, whose rounding depends on the vector ISA the
# generating host dispatches to (AVX-512: Sleef expf_u10 on 16-lane vectors with a scalar tail at n % 32, which is what
# rtm3d_amd/csrc/decode2d.hip reproduces).  Regenerating on an AVX2-only box would silently change the expected bits.
assert torch.backends.cpu.get_cpu_capability() == 'AVX512', \
    'golden vectors must be generated on an AVX-512 host (got %s)' % torch.backends.cpu.get_cpu_capability()


def make_cfg(backbone, thresh=0.4, topk=100, nconv=2):
    return NS(MODEL=NS(BACKBONE=backbone, DOWN_SAMPLE=4., OUT_CHANNELS=256, KFNs=KFNS[backbone], HEADER_NUM_CONV=nconv),
              DATASET=NS(OBJs=['Car', 'Pedestrian', 'Cyclist'], VERTEX_OFFSET_INFER=[0.75, 0.57]),
              DETECTOR=NS(SCORE_THRESH=thresh, TOPK_CANDIDATES=topk))


def ref_model(backbone, sd, thresh=0.4, topk=100, nconv=2):
    cfg = make_cfg(backbone, thresh, topk, nconv)
    bb = dla.create_model(cfg) if 'DLA' in backbone else resnet.get_pose_net(backbone.split('-')[-1], cfg)
    m = Model(cfg, bb).eval()
    if sd is not None:
        m.load_state_dict(sd)
    return m


def dets_to_arrays(dets, prefix, out):
    clses, scores, mprojs, verts, boxes = dets
    B = len(clses)
    n = np.array([0 if c is None else len(c) for c in clses], np.int32)
    out[prefix + 'n'] = n
    for b in range(B):
        if clses[b] is None:
            continue
        out['%scls_%d' % (prefix, b)] = clses[b].numpy().astype(np.int64)
        out['%sscore_%d' % (prefix, b)] = scores[b].numpy()
        out['%smproj_%d' % (prefix, b)] = mprojs[b].numpy()
        out['%sverts_%d' % (prefix, b)] = verts[b].numpy()
        out['%sbbox_%d' % (prefix, b)] = boxes[b].numpy()


def assert_tie_free(main_kf_logits, topk):
    """CPU topk tie order is implementation-defined; make sure fixtures never depend on it."""
    for i in range(main_kf_logits.shape[0]):
        hm = ref_mu.nms_hm(torch.sigmoid(main_kf_logits[i:i + 1].clone()), 3).reshape(-1)
        s, _ = torch.topk(hm, topk + 1)
        s = s[s > 0]
        assert len(torch.unique(s)) == len(s), 'tie among the top-(k+1) scores of image %d' % i


OPTIONS = {'disp': None, 'maxcor': 10, 'ftol': 2.220446049250313e-09, 'gtol': 1e-05, 'eps': 1e-08,
           'maxfun': 15000, 'maxiter': 15000, 'iprint': -1, 'maxls': 20, 'finite_diff_rel_step': None}   # utils/model_utils.py:290-291


def decode3d_to_arrays(dets, K, prefix, out):
    """Per image: the reference's optim_decode_bbox3d (kept boxes) AND the raw optimiser state (x, fun, nit) of
    EVERY detection, kept or not, obtained by calling SciPy with the reference's own aimFun / jac."""
    from oracle.decode3d_ref import COR
    K33 = K.reshape(3, 3)
    kept_total = 0
    for b in range(len(dets[0])):
        if dets[0][b] is None:
            continue
        clses, verts = dets[0][b].numpy(), dets[3][b].numpy()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pl = ref_mu.optim_decode_bbox3d(clses, verts, K.copy(), DIM_REF, list(REF_LOC))
            raw = {'x': [], 'fun': [], 'nit': []}
            for cls, UV in zip(clses, verts):
                dim = DIM_REF[cls]
                X0 = np.array([0, 1] + [dim[2], dim[0], dim[1]] + REF_LOC)
                res = minimize(ref_mu.aimFun(*(COR, K33, UV.T)), X0, method='L-BFGS-B',
                               jac=ref_mu.jac(*(COR, K33, UV.T)), options=OPTIONS)
                raw['x'].append(res.x); raw['fun'].append(res.fun); raw['nit'].append(res.nit)
        out['%sclass_%d' % (prefix, b)] = np.array(pl.get_field('class'), np.int64)
        out['%sRy_%d' % (prefix, b)] = np.asarray(pl.get_field('Ry'), np.float64)
        out['%sdimension_%d' % (prefix, b)] = np.asarray(pl.get_field('dimension'), np.float64).reshape(-1, 3)
        out['%slocation_%d' % (prefix, b)] = np.asarray(pl.get_field('location'), np.float64).reshape(-1, 3)
        out['%sraw_x_%d' % (prefix, b)] = np.array(raw['x'])
        out['%sraw_fun_%d' % (prefix, b)] = np.array(raw['fun'])
        out['%sraw_nit_%d' % (prefix, b)] = np.array(raw['nit'])
        kept_total += len(pl.get_field('class'))
    return kept_total


# (backbone, seed, heat_bias, heat_gain, B, H, W, tag): heat_gain/heat_bias chosen so that each image has ~20-30
# detections whose scores spread over 0.4 .. 0.95 (random features alone give a narrow band of peak heights)
NC1_HB, NC1_HG, NC3_HB, NC3_HG = -6.75, 3.5, -12.4, 3.5
E2E_CASES = [('DLA-34', 1, -16.9, 3.5, 2, 128, 256, 'small'), ('RESNET-18', 1, -22.8, 6.0, 2, 128, 256, 'small'),
             ('RESNET-34', 1, -14.3, 4.0, 2, 128, 256, 'small'),
             ('DLA-34', 1, -25.0, 4.5, 1, 384, 1280, 'full'), ('RESNET-18', 1, -30.0, 6.5, 1, 384, 1280, 'full'),
             # the real-KITTI letterbox shape: with IS_RECT the reference pads 1242 x 375 images to 1280 x 416 (datasets/dataset_reader.py:55-61):
             # level4 / level5 maps of 26 x 80 / 13 x 40 (H % 8 != 0 there: the 8 x 32 halo-tile kernels are not eligible)
             ('DLA-34', 1, -25.0, 4.5, 1, 416, 1280, 'kitti416'), ('RESNET-18', 1, -29.2, 6.5, 1, 416, 1280, 'kitti416'),
             # MODEL.HEADER_NUM_CONV other than the shipped 2 (models/nets/header.py:12-13: [6] + [1] * (n - 1) dilations): 1 and 3
             ('DLA-34', 1, NC1_HB, NC1_HG, 2, 128, 256, 'small_nc1', 1), ('DLA-34', 1, NC3_HB, NC3_HG, 2, 128, 256, 'small_nc3', 3)]


def gen_e2e(only_tag=None):
    for case in E2E_CASES:
        bb, seed, hb, hg, B, H, W, tag = case[:8]
        nconv = case[8] if len(case) > 8 else 2
        if only_tag is not None and tag != only_tag:
            continue
        sd = weights.synth_state_dict(bb, seed, 'trained', heat_bias=hb, heat_gain=hg, header_num_conv=nconv)
        m = ref_model(bb, sd, nconv=nconv)
        x = weights.synth_images(B, H, W, seed=1234)
        with torch.no_grad():
            dets, logits = m(x)
        assert_tie_free(logits[0], 100)
        out = {'backbone': bb, 'seed': seed, 'heat_bias': hb, 'heat_gain': hg, 'style': 'trained', 'shape': np.array([B, H, W]),
               'header_num_conv': nconv,
               'img_seed': 1234,
               # guards against drift of the numpy bit-stream that regenerates weights/images on another box
               'w_probe': sd['detect_header.main_kf_header.main_kf_head.weight'].numpy()[:, :4, 1, 1].copy(),
               'x_probe': x[0, :, :2, :8].numpy().copy()}
        if tag.startswith('small'):
            for i, name in enumerate(['main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset']):
                out['logits_' + name] = logits[i].numpy()
        else:
            out['logits_main_kf'] = logits[0].numpy()
            for i, name in enumerate(['offset_fr_main', 'main_offset', 'vertex_offset'], 1):
                out['logits_%s_s4' % name] = logits[i][:, :, ::4, ::4].numpy().copy()
            # full-resolution values of the regression heads at the detected key points
            for b in range(B):
                if dets[0][b] is None:
                    continue
                mp = dets[2][b] / 4.0
                xi, yi = mp[:, 0].floor().long(), mp[:, 1].floor().long()
                out['offs_at_det_%d' % b] = logits[1][b][:, yi, xi].numpy()
                out['moff_at_det_%d' % b] = logits[2][b][:, yi, xi].numpy()
        dets_to_arrays(dets, 'det_', out)
        K = weights.synth_intrinsics()
        kept = decode3d_to_arrays(dets, K, 'd3_', out)
        out['K'] = K
        name = 'e2e_%s_%s.npz' % (bb.lower().replace('-', ''), tag)
        np.savez_compressed(os.path.join(HERE, name), **out)
        sc = [np.round(out['det_score_%d' % b][[0, -1]], 3) for b in range(B) if out['det_n'][b]]
        print(name, 'ndet', out['det_n'], 'score range', sc, '3D kept', kept, 'logit absmax', [float(l.abs().max()) for l in logits])


def gen_planted():
    """Planted cuboids on reference-run logits (tests/golden/cases.py): the reference's own 2D decode + 3D decode,
    which keeps the planted objects (fun < 0.1) and rejects the background's natural detections."""
    gc = _load_cases()
    m = ref_model('DLA-34', None)
    for name, (fixture, _, nobj, _) in gc.PLANTED_CASES.items():
        bg = np.load(os.path.join(HERE, fixture), allow_pickle=False)
        th, tk, K, arrs, truth = gc.planted_inputs(name, bg)
        lg = [torch.from_numpy(a) for a in arrs]
        m.config.DETECTOR.SCORE_THRESH, m.config.DETECTOR.TOPK_CANDIDATES = th, tk
        assert_tie_free(lg[0], tk)
        with torch.no_grad():
            dets = m.inference([l.clone() for l in lg])
        out = {'probe': np.concatenate([a.reshape(-1)[:16] for a in arrs]), 'K': K, 'background': fixture}
        dets_to_arrays(dets, 'det_', out)
        kept = decode3d_to_arrays(dets, K, 'd3_', out)
        for b in range(len(truth)):
            kk = len(out['d3_class_%d' % b])
            assert kk >= 10, 'image %d: the reference kept only %d planted boxes' % (b, kk)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, 'ndet', out['det_n'], 'kept by the reference', [len(out['d3_class_%d' % b]) for b in range(len(truth))],
              'of', nobj, 'planted per image')


def _load_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location('golden_cases', os.path.join(HERE, 'cases.py'))
    gc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gc)
    return gc


def gen_decode2d():
    """Model.inference (models/model.py:29-75) on synthetic logits; weights are irrelevant here."""
    gc = _load_cases()
    DECODE2D_CASES, decode2d_inputs = gc.DECODE2D_CASES, gc.decode2d_inputs
    m = ref_model('RESNET-18', None)
    out = {}
    for name in DECODE2D_CASES:
        th, tk, arrs = decode2d_inputs(name)
        lg = [torch.from_numpy(a) for a in arrs]
        m.config.DETECTOR.SCORE_THRESH, m.config.DETECTOR.TOPK_CANDIDATES = th, tk
        if name != 'plateau':
            assert_tie_free(lg[0], tk)
        with torch.no_grad():
            dets = m.inference([l.clone() for l in lg])
        out[name + '_probe'] = np.concatenate([a.reshape(-1)[:16] for a in arrs])
        dets_to_arrays(dets, name + '_det_', out)
        print('decode2d', name, out[name + '_det_n'])
    np.savez_compressed(os.path.join(HERE, 'decode2d_cases.npz'), **out)


def gen_decode3d():
    """optim_decode_bbox3d (utils/model_utils.py:264-312) on synthetic key points; the raw optimiser
    state (x, fun, nit) is recorded by calling SciPy with the reference's own aimFun/jac."""
    from oracle.decode3d_ref import project_box, COR
    rng = np.random.Generator(np.random.PCG64(11))
    K = weights.synth_intrinsics()
    clses, uvs, noise_tag = [], [], []
    for noise in (0.0, 0.01, 0.3, 2.0):
        for _ in range(16):
            cls = int(rng.integers(0, 3))
            dim = np.array(DIM_REF[cls]) * rng.uniform(0.8, 1.25, 3)
            loc = np.array([rng.uniform(-12, 12), rng.uniform(0.5, 1.6), rng.uniform(6, 55)])
            ry = rng.uniform(-np.pi, np.pi)
            uv = project_box(dim, loc, ry, K) + noise * rng.standard_normal((8, 2))
            clses.append(cls); uvs.append(uv.astype(np.float32)); noise_tag.append(noise)
    clses = np.array(clses, np.int64)
    uvs = np.stack(uvs)                      # fp32, as v_projs_regress arrives from the model (detect.py:72)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pl = ref_mu.optim_decode_bbox3d(clses, uvs, K.copy(), DIM_REF, list(REF_LOC))
        raw = {'x': [], 'fun': [], 'nit': [], 'nfev': []}
        options = OPTIONS
        K33 = K.reshape(3, 3)
        for cls, UV in zip(clses, uvs):
            dim = DIM_REF[cls]
            X0 = np.array([0, 1] + [dim[2], dim[0], dim[1]] + REF_LOC)
            res = minimize(ref_mu.aimFun(*(COR, K33, UV.T)), X0, method='L-BFGS-B',
                           jac=ref_mu.jac(*(COR, K33, UV.T)), options=options)
            raw['x'].append(res.x); raw['fun'].append(res.fun); raw['nit'].append(res.nit); raw['nfev'].append(res.nfev)
    out = {'clses': clses, 'uv': uvs, 'K': K, 'dim_ref': np.array(DIM_REF), 'ref_loc': np.array(REF_LOC, np.float64),
           'noise': np.array(noise_tag),
           'out_class': np.array(pl.get_field('class'), np.int64), 'out_Ry': np.asarray(pl.get_field('Ry')),
           'out_dimension': np.asarray(pl.get_field('dimension')), 'out_location': np.asarray(pl.get_field('location')),
           'out_K': np.asarray(pl.get_field('K')),
           'raw_x': np.array(raw['x']), 'raw_fun': np.array(raw['fun']), 'raw_nit': np.array(raw['nit']),
           'raw_nfev': np.array(raw['nfev'])}
    np.savez_compressed(os.path.join(HERE, 'decode3d_cases.npz'), **out)
    print('decode3d: %d objects, kept %d, nit %d..%d' % (len(clses), len(out['out_class']), out['raw_nit'].min(), out['raw_nit'].max()))
    # empty case behaviour (:307-311)
    pl = ref_mu.optim_decode_bbox3d(np.zeros((0,), np.int64), np.zeros((0, 8, 2), np.float32), K.copy(), DIM_REF, list(REF_LOC))
    assert pl.get_field('dimension').shape == (0, 3) and pl.get_field('K').shape == (0, 9)


# ---- a large solver fixture: >= 1500 objects through SciPy with the reference's own aimFun / jac
LARGE_NOISE = [(0.0, 256), (0.01, 384), (0.03, 256), (0.1, 256), (0.3, 256), (2.0, 128)]      # (pixel noise sigma, objects)


def _large_chunk(args):
    """Worker: the reference's optim_decode_bbox3d on a chunk + the raw optimiser state of every object."""
    clses, uvs, K = args
    from oracle.decode3d_ref import COR
    K33 = K.reshape(3, 3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pl = ref_mu.optim_decode_bbox3d(clses, uvs, K.copy(), DIM_REF, list(REF_LOC))
        x, fun, nit = [], [], []
        for cls, UV in zip(clses, uvs):
            dim = DIM_REF[cls]
            X0 = np.array([0, 1] + [dim[2], dim[0], dim[1]] + REF_LOC)
            res = minimize(ref_mu.aimFun(*(COR, K33, UV.T)), X0, method='L-BFGS-B', jac=ref_mu.jac(*(COR, K33, UV.T)), options=OPTIONS)
            x.append(res.x); fun.append(res.fun); nit.append(res.nit)
    return (np.array(pl.get_field('class'), np.int64), np.asarray(pl.get_field('Ry'), np.float64),
            np.asarray(pl.get_field('dimension'), np.float64).reshape(-1, 3), np.asarray(pl.get_field('location'), np.float64).reshape(-1, 3),
            np.array(x), np.array(fun), np.array(nit))


def gen_decode3d_large():
    """1536 synthetic cuboids (exact projections + pixel noise of six levels, fp32 key points as the model hands them over) through
    the reference's optim_decode_bbox3d and, object by object, through SciPy with the reference's aimFun / jac (raw x, fun, nit):
    pins the TAIL of the device solvers (tests/test_gpu_parity.py::test_decode3d_large_fixture_*)."""
    import multiprocessing as mp
    from oracle.decode3d_ref import project_box
    rng = np.random.Generator(np.random.PCG64(4041))
    K = weights.synth_intrinsics()
    clses, uvs, noise_tag = [], [], []
    for noise, count in LARGE_NOISE:
        for _ in range(count):
            cls = int(rng.integers(0, 3))
            dim = np.array(DIM_REF[cls]) * rng.uniform(0.8, 1.25, 3)
            loc = np.array([rng.uniform(-12, 12), rng.uniform(0.5, 1.6), rng.uniform(6, 55)])
            ry = rng.uniform(-np.pi, np.pi)
            uv = project_box(dim, loc, ry, K) + noise * rng.standard_normal((8, 2))
            clses.append(cls); uvs.append(uv.astype(np.float32)); noise_tag.append(noise)
    clses, uvs = np.array(clses, np.int64), np.stack(uvs)
    n, step = len(clses), 64
    chunks = [(clses[i:i + step], uvs[i:i + step], K) for i in range(0, n, step)]
    with mp.get_context('fork').Pool(8) as pool:
        parts = pool.map(_large_chunk, chunks)
    out = {'clses': clses, 'uv': uvs, 'K': K, 'dim_ref': np.array(DIM_REF), 'ref_loc': np.array(REF_LOC, np.float64), 'noise': np.array(noise_tag),
           'out_class': np.concatenate([p[0] for p in parts]), 'out_Ry': np.concatenate([p[1] for p in parts]),
           'out_dimension': np.concatenate([p[2] for p in parts]), 'out_location': np.concatenate([p[3] for p in parts]),
           'raw_x': np.concatenate([p[4] for p in parts]), 'raw_fun': np.concatenate([p[5] for p in parts]), 'raw_nit': np.concatenate([p[6] for p in parts])}
    kept = out['raw_fun'] < 0.1
    assert kept.sum() == len(out['out_class'])
    np.savez_compressed(os.path.join(HERE, 'decode3d_large.npz'), **out)
    print('decode3d_large: %d objects, kept %d, nit %d..%d; kept per noise level %s' % (
        n, kept.sum(), out['raw_nit'].min(), out['raw_nit'].max(), {nz: int(kept[out['noise'] == nz].sum()) for nz, _ in LARGE_NOISE}))


if __name__ == '__main__':
    which = sys.argv[1:] or ['e2e', 'planted', 'decode2d', 'decode3d']
    if 'decode3d' in which:
        gen_decode3d()
    if 'decode3d_large' in which:
        gen_decode3d_large()
    if 'decode2d' in which:
        gen_decode2d()
    if 'e2e' in which:
        gen_e2e()
    for w in which:
        if w.startswith('e2e:'):
            gen_e2e(w.split(':', 1)[1])                 # only the cases of one tag, e.g. e2e:kitti416
    if 'planted' in which:
        gen_planted()
