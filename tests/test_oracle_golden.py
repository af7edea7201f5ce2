"""CPU: pin the oracle (oracle/) against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  Bit-exact for the forward/2D decode (same PyTorch CPU kernels),
1e-9 for the SciPy-driven 3D decode."""
import numpy as np
import pytest
import torch

from oracle import rtm3d_ref, decode3d_ref
from rtm3d_amd import weights
from tests.golden.cases import DECODE2D_CASES, decode2d_inputs, PLANTED_CASES, planted_inputs, DIM_REF
from tests.util import load_golden, dets_from_golden, to_np, canon_dets

E2E = ['e2e_dla34_small.npz', 'e2e_resnet18_small.npz', 'e2e_resnet34_small.npz', 'e2e_dla34_full.npz', 'e2e_resnet18_full.npz',
       'e2e_dla34_kitti416.npz', 'e2e_resnet18_kitti416.npz']
# MODEL.HEADER_NUM_CONV = 1 and 3 (models/nets/header.py:12-13), run through the reference like the others
E2E_NC = ['e2e_dla34_small_nc1.npz', 'e2e_dla34_small_nc3.npz']


@pytest.mark.parametrize('fname', E2E + E2E_NC)
def test_oracle_forward_matches_reference(fname):
    g = load_golden(fname)
    bb = str(g['backbone'])
    B, H, W = [int(v) for v in g['shape']]
    sd = weights.synth_state_dict(bb, int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']),
                                  header_num_conv=int(g['header_num_conv']) if 'header_num_conv' in g else 2)
    x = weights.synth_images(B, H, W, seed=int(g['img_seed']))
    # the seeded generators must reproduce the tensors the reference was run on
    np.testing.assert_array_equal(sd['detect_header.main_kf_header.main_kf_head.weight'].numpy()[:, :4, 1, 1], g['w_probe'])
    np.testing.assert_array_equal(x[0, :, :2, :8].numpy(), g['x_probe'])
    torch.set_num_threads(8)
    dets, logits = rtm3d_ref.model_forward(x, sd, bb)
    # conv results may differ in the last bits with the thread count / oneDNN blocking of the host
    np.testing.assert_allclose(logits[0].numpy(), g['logits_main_kf'], rtol=0, atol=2e-5)
    if 'logits_offset_fr_main' in g:
        for i, name in enumerate(['offset_fr_main', 'main_offset', 'vertex_offset'], 1):
            np.testing.assert_allclose(logits[i].numpy(), g['logits_' + name], rtol=0, atol=2e-5)
    else:
        for i, name in enumerate(['offset_fr_main', 'main_offset', 'vertex_offset'], 1):
            np.testing.assert_allclose(logits[i][:, :, ::4, ::4].numpy(), g['logits_%s_s4' % name], rtol=0, atol=2e-5)
    n = g['det_n']
    for b in range(B):
        if n[b] == 0:
            assert dets[0][b] is None
            continue
        ref = dets_from_golden(g, 'det_', b)
        np.testing.assert_array_equal(dets[0][b].numpy(), ref[0])           # classes: exact
        np.testing.assert_allclose(dets[1][b].numpy(), ref[1], atol=1e-5)
        for k in (2, 3, 4):
            np.testing.assert_allclose(dets[k][b].numpy(), ref[k], atol=2e-4)
    # 3D decode of image 0's reference detections: every detection's raw optimiser state, kept or not
    if n[0]:
        ref = dets_from_golden(g, 'det_', 0)
        _check_decode3d(g, 0, ref[0], ref[3], g['K'])


def _check_decode3d(g, b, clses, verts, K):
    res, raw = decode3d_ref.optim_decode_bbox3d(clses, verts, K, DIM_REF, [0, -0.5, 20], return_raw=True)
    np.testing.assert_array_equal(np.array(res['class'], np.int64), g['d3_class_%d' % b])
    np.testing.assert_array_equal(raw['nit'], g['d3_raw_nit_%d' % b])
    np.testing.assert_allclose(raw['x'], g['d3_raw_x_%d' % b], rtol=0, atol=1e-9)
    np.testing.assert_allclose(raw['fun'], g['d3_raw_fun_%d' % b], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(res['location'], g['d3_location_%d' % b], atol=1e-9)
    np.testing.assert_allclose(res['dimension'], g['d3_dimension_%d' % b], atol=1e-9)
    np.testing.assert_allclose(res['Ry'], g['d3_Ry_%d' % b], atol=1e-9)


@pytest.mark.parametrize('name', sorted(PLANTED_CASES))
def test_oracle_planted_pipeline_matches_reference(name):
    """Planted cuboids on reference-run logits: the oracle's 2D decode is bit-exact and its 3D decode keeps exactly
    the objects the reference kept (>= 10 per image), with the same boxes."""
    g = load_golden(name + '.npz')
    th, tk, K, arrs, truth = planted_inputs(name, load_golden(PLANTED_CASES[name][0]))
    np.testing.assert_array_equal(np.concatenate([a.reshape(-1)[:16] for a in arrs]), g['probe'])
    dets = rtm3d_ref.inference([torch.from_numpy(a) for a in arrs], th, tk, 4.0)
    for b in range(len(g['det_n'])):
        got = [to_np(d[b]) for d in dets]
        for a, r in zip(got, dets_from_golden(g, 'det_', b)):
            np.testing.assert_array_equal(a, r)
        assert len(g['d3_class_%d' % b]) >= 10 and (g['d3_raw_fun_%d' % b] >= 0.1).any()     # kept AND rejected objects
        _check_decode3d(g, b, got[0], got[3], K)


@pytest.mark.parametrize('name', DECODE2D_CASES)
def test_oracle_decode2d_bit_exact(name):
    g = load_golden('decode2d_cases.npz')
    th, tk, arrs = decode2d_inputs(name)
    np.testing.assert_array_equal(np.concatenate([a.reshape(-1)[:16] for a in arrs]), g[name + '_probe'])
    dets = rtm3d_ref.inference([torch.from_numpy(a) for a in arrs], th, tk, 4.0)
    n = g[name + '_det_n']
    for b in range(len(n)):
        if n[b] == 0:
            assert dets[0][b] is None
            continue
        got = [to_np(d[b]) for d in dets]
        ref = dets_from_golden(g, name + '_det_', b)
        if name == 'plateau':     # tie order of torch.topk is implementation-defined: compare canonically
            got, ref = canon_dets(*got), canon_dets(*ref)
        for a, r in zip(got, ref):
            np.testing.assert_array_equal(a, r)


def test_oracle_decode3d_matches_reference():
    g = load_golden('decode3d_cases.npz')
    res, raw = decode3d_ref.optim_decode_bbox3d(g['clses'], g['uv'], g['K'], g['dim_ref'].tolist(),
                                                g['ref_loc'].tolist(), return_raw=True)
    np.testing.assert_array_equal(np.array(res['class']), g['out_class'])
    np.testing.assert_array_equal(raw['nit'], g['raw_nit'])
    np.testing.assert_allclose(raw['x'], g['raw_x'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(raw['fun'], g['raw_fun'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res['Ry'], g['out_Ry'], atol=1e-9)
    np.testing.assert_allclose(res['dimension'], g['out_dimension'], atol=1e-9)
    np.testing.assert_allclose(res['location'], g['out_location'], atol=1e-9)
    np.testing.assert_array_equal(res['K'], g['out_K'])
    # empty input (utils/model_utils.py:307-311)
    e = decode3d_ref.optim_decode_bbox3d(np.zeros((0,), np.int64), np.zeros((0, 8, 2)), g['K'], g['dim_ref'].tolist(), [0, -0.5, 20])
    assert e['dimension'].shape == (0, 3) and e['K'].shape == (0, 9) and e['class'] == []
