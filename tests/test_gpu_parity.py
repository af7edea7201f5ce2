"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes), against the golden
vectors of the real reference and against the CPU oracle on fresh seeded inputs.

Bars: decode indices / classes bit-exact; decode float outputs bit-exact (same fp32 operation order
and the ATen sigmoid reproduced); 3D boxes within 1e-4 (north_star); network logits (fp16 storage,
fp32 accumulation) within 2 x the measured error of the fp32 reference (LOGIT_RTOL below); in the fp32
verification mode within 2e-5 of scale, identical detections."""
import ctypes

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rtm3d_ref, decode3d_ref          # noqa: E402  (the checker)
import rtm3d_amd                                     # noqa: E402
from rtm3d_amd import weights, _lib                  # noqa: E402
from tests.golden.cases import DECODE2D_CASES, decode2d_inputs, PLANTED_CASES, planted_inputs   # noqa: E402
from tests.util import load_golden, dets_from_golden, canon_dets, to_np, record_measurement, pack_records_reference   # noqa: E402

# fp16 activations/weights, fp32 accumulation, ~45 layers: |err| <= tol * max(1, max|ref|) per tensor, tol = 2 x the
# largest error MEASURED on the MI355X over every fixture / backbone for that stage (profiles/r02_logit_error.json, written
# by these tests through tests/util.record_measurement): logits 0.0042, fused map z 0.0063, backbone features 0.0014.
LOGIT_RTOL = 0.010
Z_RTOL = 0.009
FEAT_RTOL = 0.003
# Full-size maps (round 4): the fused map z = z0 + sum u * softmax_HW(u) has isolated pixels where a PEAKED spatial softmax multiplies
# the fp16 rounding of u (exp(u - max) moves by ulp(u) ~ 1-3 %): measured 0.0119 at ONE pixel of the 352 x 1216 DLA-34 map with the
# 99.9th percentile at 3.8e-4 and the fp32 verification mode at 2.4e-5 on the same map (tools/gpu_z_odd_shape.py) - storage rounding,
# not wiring.  Bars = 2 x measured.
Z_PEAK_RTOL = 0.024
Z_P999_RTOL = 8e-4
Z_FP32_RTOL = 5e-5
HM_RTOL = 0.0055       # heat-map logits alone (measured 0.0028): decides which reference detections are safely above the threshold
VERT_TOL_PX = 0.25    # vertices of matched detections: 16 regression channels x stride 4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    _lib.load()
    return torch.device('cuda', 0)


def make_model(bb, sd=None, thresh=0.4, topk=100, nconv=2):
    cfg = rtm3d_amd.kitti_config(bb)
    cfg.DETECTOR.SCORE_THRESH, cfg.DETECTOR.TOPK_CANDIDATES = thresh, topk
    cfg.MODEL.HEADER_NUM_CONV = nconv
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    if sd is not None:
        m.load_state_dict(sd)
    return m


# ------------------------------------------------------------------------------ 2D decode
@pytest.mark.parametrize('name', DECODE2D_CASES)
def test_decode2d_golden_bit_exact(dev, name):
    g = load_golden('decode2d_cases.npz')
    th, tk, arrs = decode2d_inputs(name)
    m = make_model('RESNET-18', None, th, tk)
    d = m.inference([torch.from_numpy(a).to(dev) for a in arrs])
    n = g[name + '_det_n']
    for b in range(len(n)):
        if n[b] == 0:
            assert d[0][b] is None
            continue
        got = [to_np(x[b]) for x in d]
        ref = dets_from_golden(g, name + '_det_', b)
        assert len(got[0]) == n[b]
        if name == 'plateau':          # order inside a score tie is implementation-defined in the reference
            got, ref = canon_dets(*got), canon_dets(*ref)
        for a, r in zip(got, ref):
            np.testing.assert_array_equal(a, r)


def test_decode2d_random_vs_oracle(dev):
    rng = np.random.Generator(np.random.PCG64(99))
    for (B, H, W, th, tk) in [(3, 96, 320, 0.4, 100), (2, 40, 72, 0.3, 50), (1, 7, 9, 0.4, 100), (2, 96, 320, 0.05, 256)]:
        lg = [torch.from_numpy((rng.standard_normal((B, c, H, W)) * s + o).astype(np.float32))
              for c, s, o in ((3, 1.5, -2.5), (16, 3, 0), (2, 2, 0), (2, 1, 0))]
        m = make_model('RESNET-18', None, th, tk)
        d = m.inference([t.to(dev) for t in lg])
        r = rtm3d_ref.inference(lg, th, tk, 4.0)
        for b in range(B):
            if r[0][b] is None:
                assert d[0][b] is None
                continue
            # skip images whose top-(k+1) scores tie (reference order undefined there)
            sc = r[1][b].numpy()
            if len(np.unique(sc)) != len(sc):
                continue
            for k in range(5):
                np.testing.assert_array_equal(to_np(d[k][b]), to_np(r[k][b]))


def test_decode2d_topk_determinism_with_ties(dev):
    """All-equal heat map: every pixel is a plateau member; order must be ascending flat index."""
    H, W = 8, 16
    lg = [torch.full((1, 3, H, W), 2.0), torch.zeros(1, 16, H, W), torch.zeros(1, 2, H, W), torch.zeros(1, 2, H, W)]
    m = make_model('RESNET-18', None, 0.4, 100)
    d = m.inference([t.to(dev) for t in lg])
    assert len(d[0][0]) == 100
    x = (d[2][0][:, 0].cpu().numpy() / 4.0 - 0.5).round().astype(int)
    y = (d[2][0][:, 1].cpu().numpy() / 4.0 - 0.5).round().astype(int)
    flat = d[0][0].cpu().numpy() * H * W + y * W + x
    np.testing.assert_array_equal(flat, np.arange(100))


# ------------------------------------------------------------------------------ 3D decode
def test_decode3d_golden(dev):
    g = load_golden('decode3d_cases.npz')
    x, fun, nit, st = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'])
    kept_ref = g['raw_fun'] < 0.1
    np.testing.assert_array_equal(fun < 0.1, kept_ref)
    np.testing.assert_allclose(x[kept_ref], g['raw_x'][kept_ref], rtol=0, atol=1e-6)
    np.testing.assert_allclose(x, g['raw_x'], rtol=0, atol=1e-4)          # north_star tolerance, incl. rejected objects
    np.testing.assert_allclose(fun, g['raw_fun'], rtol=1e-5, atol=1e-8)   # (a 2.7e-6 minimum reached one iteration apart: 1.5e-9)
    out = rtm3d_amd.model_utils.optim_decode_bbox3d(g['clses'], g['uv'], g['K'], g['dim_ref'].tolist(), g['ref_loc'].tolist())
    assert out.get_field('class') == g['out_class'].tolist()
    np.testing.assert_allclose(out.get_field('Ry'), g['out_Ry'], atol=1e-4)
    np.testing.assert_allclose(out.get_field('dimension'), g['out_dimension'], atol=1e-4)
    np.testing.assert_allclose(out.get_field('location'), g['out_location'], atol=1e-4)
    np.testing.assert_array_equal(out.get_field('K'), g['out_K'])
    e = rtm3d_amd.model_utils.optim_decode_bbox3d(np.zeros((0,), np.int64), np.zeros((0, 8, 2), np.float32), g['K'], g['dim_ref'].tolist(), [0, -0.5, 20])
    assert e.get_field('dimension').shape == (0, 3) and e.get_field('K').shape == (0, 9) and e.get_field('class') == []


@pytest.mark.parametrize('form', ['published', 'direct'])
def test_decode3d_wave_kernel_equals_scalar_kernel(dev, form):
    """The wave-cooperative solvers perform the same fp64 operations in the same order as the one-lane-per-object forms of
    lbfgsb.h (published: lb_minimize(direct = 0) = rtm3d_decode3d_reference_form; direct: rtm3d_decode3d_scalar): bit-identical."""
    g = load_golden('decode3d_cases.npz')
    a = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], form=form)
    b = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], scalar_kernel=form == 'direct',
                                          reference_form=form == 'published')
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    if form == 'published':          # ... and it is what the facade runs when nothing is said
        assert rtm3d_amd.model_utils.DEFAULT_SOLVER_FORM == 'published'
        c = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'])
        for u, v in zip(a, c):
            np.testing.assert_array_equal(u, v)


def test_decode3d_direct_form_vs_published_form(dev):
    """The opt-in direct form computes the search direction by the two-loop recursion; the published subspace step (formk / subsm /
    formt, the form SciPy runs and the product's default) on the same objects: identical keep / reject decisions, kept boxes
    within 1e-6 (bar 1e-4); objects the reference rejects are not compared in x (their path is chaotic in either form) but end
    at the same objective value."""
    g = load_golden('decode3d_cases.npz')
    a = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], form='direct')
    b = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], reference_form=True)
    kept = b[1] < 0.1
    np.testing.assert_array_equal(a[1] < 0.1, kept)
    np.testing.assert_array_equal(kept, g['raw_fun'] < 0.1)
    np.testing.assert_allclose(a[0][kept], b[0][kept], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=1e-8)
    record_measurement('decode3d_product_vs_published_form', 'decode3d_cases',
                       {'kept_x_linf': float(np.abs(a[0] - b[0])[kept].max()), 'all_x_linf': float(np.abs(a[0] - b[0]).max()),
                        'nit_mean_product': float(a[2].mean()), 'nit_mean_published': float(b[2].mean())})
    # the published form on the device reproduces the host build of the same header bit for bit (no contraction, same order)
    np.testing.assert_allclose(b[0][kept], g['raw_x'][kept], rtol=0, atol=1e-7)


def _solve_through_slots(g, dev, form, topk=96):
    """The 1536 fixture objects through the PRODUCT entry (rtm3d_decode3d_slots: the fixed-size slots decode2d writes, one K per
    image) with the given search-direction form."""
    from rtm3d_amd.model import Detections
    from rtm3d_amd.model_utils import decode3d_slots
    N = len(g['clses'])
    assert N % topk == 0 and np.asarray(g['K']).size == 9
    B = N // topk
    det = Detections(B, topk, dev)
    det.n[:] = topk
    det.cls[:] = torch.as_tensor(np.asarray(g['clses']).astype(np.int64), device=dev)
    det.verts[:] = torch.as_tensor(np.asarray(g['uv'], np.float32).reshape(N, 8, 2), device=dev)
    Kd = torch.as_tensor(np.tile(np.asarray(g['K'], np.float64).reshape(1, 9), (B, 1)), device=dev)
    bx = decode3d_slots(det, Kd, np.asarray(g['dim_ref'], np.float64), np.asarray(g['ref_loc'], np.float64), form=form)
    torch.cuda.synchronize()
    assert int((bx.status < 0).sum()) == 0
    return bx.x.cpu().numpy(), bx.fun.cpu().numpy(), bx.nit.cpu().numpy()


# the two forms of the search direction (rtm3d_amd/csrc/lbfgsb.h) and the kernels that run them
PUBLISHED_FORM = ('wave', 'reference_form', 'slots_published')      # formk / subsm (what SciPy runs), THE DEFAULT: rtm3d_decode3d as the facade calls it, the scalar cross-check, the slot entry
DIRECT_FORM = ('wave_direct', 'scalar', 'slots_direct')             # two-loop recursion (opt-in): rtm3d_decode3d(form = direct), its scalar twin, the slot entry with form = 0


@pytest.mark.parametrize('kernel', DIRECT_FORM + PUBLISHED_FORM)
def test_decode3d_large_fixture(dev, kernel):
    """The TAIL of the device solvers, pinned.  1536 objects the reference solved (SciPy through its own aimFun / jac;
    tests/golden/decode3d_large.npz, 876 kept, six noise levels).  Keep / reject identical for every kernel.
    PUBLISHED form (the arithmetic utils/model_utils.py:295-296 runs through SciPy; the product's default since round 6): EVERY
    kept box within north_star's 1e-4 (measured 1.3e-5).  DIRECT form (opt-in, two-loop search direction): >= 99.5 % of the kept
    boxes within 1e-4, p99 <= 1e-5 - an object in ~1000 ends an iteration apart from SciPy (measured: all 876 within 2.7e-5 on
    this fixture, 110 of 111 on the bench's planted boxes)."""
    from tests.util import solver_tail_stats
    g = load_golden('decode3d_large.npz')
    if kernel.startswith('slots_'):
        x, fun, nit = _solve_through_slots(g, dev, kernel[len('slots_'):])
    else:
        x, fun, nit, _ = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'],
                                                           scalar_kernel=kernel == 'scalar', reference_form=kernel == 'reference_form',
                                                           form='direct' if kernel == 'wave_direct' else None)
    s = solver_tail_stats(x, fun, g)
    record_measurement('decode3d_large_fixture', kernel, s)
    assert s['n'] >= 1500 and s['keep_mismatch'] == 0 and s['kept'] >= 800, s
    if kernel in PUBLISHED_FORM:
        assert s['within_1e-4'] == 1.0 and s['p99'] <= 1e-5, s
    else:
        assert s['within_1e-4'] >= 0.995 and s['p99'] <= 1e-5, s
    if kernel == 'wave':             # the drop-in entry (utils/model_utils.py:264-312) runs the default form: every box within 1e-4
        out = rtm3d_amd.model_utils.optim_decode_bbox3d(g['clses'], g['uv'], g['K'], g['dim_ref'].tolist(), g['ref_loc'].tolist())
        assert out.get_field('class') == g['out_class'].tolist()
        for f in ('dimension', 'location'):
            assert np.abs(np.asarray(out.get_field(f)) - g['out_' + f]).max() <= 1e-4, f
        dr = np.abs(np.asarray(out.get_field('Ry')) - g['out_Ry'])
        assert np.minimum(dr, 2 * np.pi - dr).max() <= 1e-4
    if kernel.startswith('slots_'):
        # the slot entry runs the same arithmetic as the flat cross-check entry of its form: bit for bit
        ref = rtm3d_amd.model_utils.solve_boxes(g['clses'], g['uv'], g['K'], g['dim_ref'], g['ref_loc'], scalar_kernel=kernel == 'slots_direct',
                                                reference_form=kernel == 'slots_published')
        np.testing.assert_array_equal(nit, ref[2])
        np.testing.assert_array_equal(x, ref[0])
        np.testing.assert_array_equal(fun, ref[1])


def test_decode3d_slots_refuses_an_unknown_form(dev):
    from rtm3d_amd.model import Detections
    from rtm3d_amd.model_utils import decode3d_slots
    det = Detections(1, 4, dev)
    K = torch.as_tensor(weights.synth_intrinsics().reshape(1, 9), device=dev)
    with pytest.raises(ValueError, match='solver form'):
        decode3d_slots(det, K, rtm3d_amd.kitti_config().DETECTOR.dim_ref, form='scipy')
    lib = _lib.load()
    z = torch.zeros(64, dtype=torch.float64, device=dev)
    rc = lib.rtm3d_decode3d_slots(ctypes.c_void_p(0), 1, 4, det.n.data_ptr(), det.cls.data_ptr(), det.verts.data_ptr(), K.data_ptr(),
                                  z.data_ptr(), 3, z.data_ptr(), z.data_ptr(), z.data_ptr(), det.n.data_ptr(), det.n.data_ptr(), 7)
    assert rc != 0 and b'solver form' in lib.rtm3d_last_error()
    with pytest.raises(ValueError, match='solver form'):
        from rtm3d_amd.pipeline import Detect3DPipeline
        Detect3DPipeline(make_model('DLA-34', weights.synth_state_dict('DLA-34', 1, 'trained')), 1, dev, solver_form='lbfgs')


def test_decode3d_random_vs_scipy(dev):
    """Fresh objects: same answer as the SciPy-driven oracle within 1e-4 (kept objects)."""
    rng = np.random.Generator(np.random.PCG64(2024))
    K = weights.synth_intrinsics()
    dim_ref = rtm3d_amd.kitti_config().DETECTOR.dim_ref
    clses, uvs = [], []
    for i in range(96):
        cls = int(rng.integers(0, 3))
        dim = np.array(dim_ref[cls]) * rng.uniform(0.8, 1.25, 3)
        loc = np.array([rng.uniform(-12, 12), rng.uniform(0.5, 1.6), rng.uniform(6, 55)])
        uv = decode3d_ref.project_box(dim, loc, rng.uniform(-np.pi, np.pi), K) + rng.choice([0.0, 0.02, 0.1]) * rng.standard_normal((8, 2))
        clses.append(cls); uvs.append(uv.astype(np.float32))
    clses, uvs = np.array(clses), np.stack(uvs)
    x, fun, _, _ = rtm3d_amd.model_utils.solve_boxes(clses, uvs, K, dim_ref, [0, -0.5, 20])
    _, raw = decode3d_ref.optim_decode_bbox3d(clses, uvs, K, dim_ref, [0, -0.5, 20], return_raw=True)
    np.testing.assert_array_equal(fun < 0.1, raw['kept'])
    np.testing.assert_allclose(x[raw['kept']], raw['x'][raw['kept']], rtol=0, atol=1e-4)


# ------------------------------------------------------------------------------ network
def _rel_err(got, ref):
    """max |got - ref| / max(1, max |ref|)"""
    return float(np.abs(got - ref).max() / max(1.0, float(np.abs(ref).max())))


# (the kitti416 pair: the real-KITTI letterbox shape 1 x 3 x 416 x 1280, datasets/dataset_reader.py:55-61 - level4 / level5 maps of
# 26 x 80 / 13 x 40, where the 8 x 32 halo-tile kernels are not eligible; every test that takes this list runs it: fp16 path,
# fp32 verification mode, peaks-only regression heads)
E2E = ['e2e_dla34_small.npz', 'e2e_resnet18_small.npz', 'e2e_resnet34_small.npz', 'e2e_dla34_full.npz', 'e2e_resnet18_full.npz',
       'e2e_dla34_kitti416.npz', 'e2e_resnet18_kitti416.npz']
# MODEL.HEADER_NUM_CONV = 1 and 3 (models/nets/header.py:12-13: one dilation-6 conv + n - 1 dilation-1 convs per branch), reference-run
E2E_NC = ['e2e_dla34_small_nc1.npz', 'e2e_dla34_small_nc3.npz']


@pytest.mark.parametrize('fname', E2E + E2E_NC)
def test_forward_logits_vs_reference_golden(dev, fname):
    g = load_golden(fname)
    bb = str(g['backbone'])
    B, H, W = [int(v) for v in g['shape']]
    nconv = int(g['header_num_conv']) if 'header_num_conv' in g else 2
    sd = weights.synth_state_dict(bb, int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']),
                                  header_num_conv=nconv)
    x = weights.synth_images(B, H, W, seed=int(g['img_seed']))
    m = make_model(bb, sd, nconv=nconv)
    (clses, scores, mprojs, verts, boxes), logits = m(x.to(dev))
    ref0 = g['logits_main_kf']
    tol = HM_RTOL * max(1.0, np.abs(ref0).max())
    errs = {'main_kf': _rel_err(logits[0].cpu().numpy(), ref0)}
    for i, name in enumerate(['offset_fr_main', 'main_offset', 'vertex_offset'], 1):
        if 'logits_' + name in g:
            ref = g['logits_' + name]; got = logits[i].cpu().numpy()
        else:
            ref = g['logits_%s_s4' % name]; got = logits[i][:, :, ::4, ::4].cpu().numpy()
        errs[name] = _rel_err(got, ref)
    record_measurement('logits_vs_reference_golden', fname, errs)
    for name, e in errs.items():
        assert e <= LOGIT_RTOL, (name, e)
    # detections: every reference detection whose heat-map logit is further than the logit tolerance from the
    # score threshold must be found at the same (class, y, x), with vertices within VERT_TOL_PX
    thr_logit = float(np.log(0.4 / 0.6))
    n = g['det_n']
    checked, vmax = 0, 0.0
    for b in range(B):
        if n[b] == 0:
            continue
        rc, rs, rm, rv, _ = dets_from_golden(g, 'det_', b)
        margin = np.abs(np.log(rs.astype(np.float64) / (1.0 - rs.astype(np.float64))) - thr_logit)
        sure = margin > tol
        if clses[b] is None:
            assert not sure.any()
            continue
        got = {(int(c), int(mx // 4), int(my // 4)): v for c, (mx, my), v in
               zip(clses[b].cpu().numpy(), mprojs[b].cpu().numpy(), verts[b].cpu().numpy())}
        for c, s, mp, v, ok in zip(rc, rs, rm, rv, sure):
            if not ok:
                continue
            key = (int(c), int(mp[0] // 4), int(mp[1] // 4))
            assert key in got, (key, s)
            vmax = max(vmax, float(np.abs(got[key] - v).max()))
            checked += 1
    record_measurement('e2e_detections_vs_reference_golden', fname, {'matched': checked, 'reference_detections': int(n.sum()),
                                                                     'vertex_linf_px': vmax})
    assert vmax < VERT_TOL_PX, vmax
    assert checked >= 12 * B, checked       # the fixtures bite: ~20 detections per image with scores 0.4 .. 0.95


def test_sparse_heads_refuse_other_header_depths(dev):
    """The peaks-only regression heads are a patch plan of exactly d6 + d1 + logit conv (plan.build_peak_plan): with
    MODEL.HEADER_NUM_CONV = 3 the call surface says so instead of evaluating the wrong receptive field."""
    m = make_model('DLA-34', weights.synth_state_dict('DLA-34', 1, 'trained', header_num_conv=3), nconv=3)
    x = weights.synth_images(1, 64, 128, seed=3).to(dev)
    with pytest.raises(NotImplementedError, match='HEADER_NUM_CONV'):
        m.detect(x)


def test_pipeline_with_three_header_convs(dev):
    """MODEL.HEADER_NUM_CONV = 3 through the whole device pipeline (forward -> decode2d -> decode3d -> records): the pipelined step
    gives the same records as the step-by-step calls, and the detections are those of the reference-run fixture's weights."""
    from rtm3d_amd.pipeline import Detect3DPipeline
    from rtm3d_amd import distributed as rdist
    g = load_golden('e2e_dla34_small_nc3.npz')
    B, H, W = [int(v) for v in g['shape']]
    sd = weights.synth_state_dict('DLA-34', int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']),
                                  header_num_conv=3)
    m = make_model('DLA-34', sd, nconv=3)
    x = weights.synth_images(B, H, W, seed=int(g['img_seed'])).to(dev)
    K = torch.as_tensor(np.tile(g['K'], (B, 1)), dtype=torch.float64, device=dev)
    det, boxes, _ = m.detect3d(x, K)
    n = det.n.cpu().numpy()
    assert abs(int(n.sum()) - int(g['det_n'].sum())) <= 4          # up to fp16 flips at the score threshold
    rec_ref = rdist.pack_records(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, det.topk, boxes).clone()
    pipe = Detect3DPipeline(m, B, dev, gather=False)
    k = pipe.submit(x, K)
    rec = pipe.results(k, copy=True)
    pipe.drain()
    assert torch.equal(rec, rec_ref)


@pytest.mark.parametrize('fname', E2E)
def test_sparse_heads_detections_vs_reference_golden(dev, fname):
    """Model.detect (peaks-only regression heads) against the REFERENCE's own detections of the reference-run e2e fixtures: as
    for the dense path, every reference detection whose heat-map logit is further than the logit tolerance from the threshold is
    found at the same (class, y, x) with its vertices within VERT_TOL_PX."""
    g = load_golden(fname)
    bb = str(g['backbone'])
    B, H, W = [int(v) for v in g['shape']]
    sd = weights.synth_state_dict(bb, int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']))
    x = weights.synth_images(B, H, W, seed=int(g['img_seed']))
    m = make_model(bb, sd)
    clses, scores, mprojs, verts, boxes = m.detect(x.to(dev))
    tol = HM_RTOL * max(1.0, np.abs(g['logits_main_kf']).max())
    thr_logit = float(np.log(0.4 / 0.6))
    n = g['det_n']
    checked, vmax = 0, 0.0
    for b in range(B):
        if n[b] == 0:
            continue
        rc, rs, rm, rv, _ = dets_from_golden(g, 'det_', b)
        margin = np.abs(np.log(rs.astype(np.float64) / (1.0 - rs.astype(np.float64))) - thr_logit)
        sure = margin > tol
        if clses[b] is None:
            assert not sure.any()
            continue
        got = {(int(c), int(mx // 4), int(my // 4)): v for c, (mx, my), v in
               zip(clses[b].cpu().numpy(), mprojs[b].cpu().numpy(), verts[b].cpu().numpy())}
        for c, s_, mp, v, ok in zip(rc, rs, rm, rv, sure):
            if not ok:
                continue
            key = (int(c), int(mp[0] // 4), int(mp[1] // 4))
            assert key in got, (key, s_)
            vmax = max(vmax, float(np.abs(got[key] - v).max()))
            checked += 1
    record_measurement('sparse_heads_detections_vs_reference_golden', fname, {'matched': checked, 'reference_detections': int(n.sum()),
                                                                              'vertex_linf_px': vmax})
    assert vmax < VERT_TOL_PX, vmax
    assert checked >= 12 * B, checked


@pytest.mark.parametrize('bb,shape', [('DLA-34', (3, 96, 160)), ('RESNET-18', (3, 96, 160)), ('DLA-34', (2, 352, 1216)), ('RESNET-18', (1, 352, 1216)),
                                      ('DLA-34', (1, 416, 1280)), ('RESNET-18', (8, 384, 1280))])
def test_forward_stages_vs_oracle(dev, bb, shape):
    """Fresh seed: backbone features, fused map and logits against the oracle - batch 3 on a non-square small input; an ODD full-size
    shape (352 x 1216: maps of 88 x 304 ... 11 x 38, no level takes a halo-tile kernel except by accident of divisibility) and the
    real-KITTI letterbox shape 416 x 1280; BASELINE config[1]'s shape (ResNet-18, bs=8, full size), where the neck
    fold of level 4 is active with a residual 3x3 conv as the producer of the space-to-depth feature copy."""
    B, H, W = shape
    sd = weights.synth_state_dict(bb, 11, 'trained', heat_bias=-3.0)
    x = weights.synth_images(B, H, W, seed=77)
    m = make_model(bb, sd)
    logits = m.forward_logits(x.to(dev))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _, lref, st = rtm3d_ref.model_forward(x, sd, bb, return_stages=True)
    plan = m._plan_for(B, H, W, dev)
    errs = {}
    for i in range(4):
        errs['feat%d' % i] = _rel_err(plan.download(plan.plan.named['feat%d' % i]), st['feats'][i].numpy())
    zgot, zref = plan.download(plan.plan.named['z']), st['z'].numpy()
    errs['z'] = _rel_err(zgot, zref)
    errs['z_p999'] = float(np.percentile(np.abs(zgot - zref), 99.9) / max(1.0, float(np.abs(zref).max())))
    for name, a, b in zip(['main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset'], logits, lref):
        errs[name] = _rel_err(a.cpu().numpy(), b.numpy())
    full = H * W > 100000
    if full:
        # the same plan in the fp32 verification mode: the wiring on this shape is the reference's function
        m.forward_logits_fp32(x.to(dev))
        errs['z_fp32_mode'] = _rel_err(m._verify[1].fetch('z').cpu().numpy(), zref)
        m.release_verify()
    record_measurement('stages_vs_oracle', '%s_%dx%dx%d' % ((bb,) + tuple(shape)), errs)
    for name, e in errs.items():
        bar = (FEAT_RTOL if name.startswith('feat') else Z_P999_RTOL if name == 'z_p999' else Z_FP32_RTOL if name == 'z_fp32_mode'
               else (Z_PEAK_RTOL if full else Z_RTOL) if name == 'z' else LOGIT_RTOL)
        assert e <= bar, (name, e, bar)


def test_neck_up_fold_vs_oracle(dev):
    """RealizedPlan._neck_up_folds (round 4): the neck's composed proj3 + head2 1x1 folded INTO the transposed conv kfpn_up3 (the
    1x1's share of the backbone feature as one more K-step per sub-pixel phase, read from a space-to-depth copy that the fused
    level-2 tail writes): fused map z and logits against the oracle with the fold on and off - the fold must really have been
    taken (one launch fewer, kernel names) and both forms meet the same bars."""
    from rtm3d_amd import plan as plan_mod
    bb, B, H, W = 'DLA-34', 2, 384, 1280
    sd = weights.synth_state_dict(bb, 5, 'trained', heat_bias=-3.0)
    x = weights.synth_images(B, H, W, seed=21)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _, lref, st = rtm3d_ref.model_forward(x, sd, bb, return_stages=True)
    res = {}
    v2_min = plan_mod.V2_MIN_TILES
    for fold in (True, False):
        plan_mod.FOLD_NECK_UP = fold
        plan_mod.V2_MIN_TILES = 8          # (the fold needs the 256-pixel kernel; at B = 2 the two coarser levels have 15-60 tiles: take it anyway)
        try:
            m = make_model(bb, sd)
            logits = m.forward_logits(x.to(dev))
            plan = m._plan_for(B, H, W, dev)
            names = plan.op_names
            for lvl in (3, 4, 5):
                assert any(n.startswith('kfpn_up%d+kfpn_proj%d' % (lvl, lvl)) for n in names) == fold, names
            assert any('+s2d' in n for n in names) == fold, names
            z = plan.download(plan.plan.named['z'])
            # (with the fold the level-2 feature exists only as its space-to-depth copy: download() gathers the phases back)
            assert bool(plan._s2d_only) == fold
            errs = {'feat0': _rel_err(plan.download(plan.plan.named['feat0']), st['feats'][0].numpy()),
                    'feat1': _rel_err(plan.download(plan.plan.named['feat1']), st['feats'][1].numpy()),
                    'z': _rel_err(z, st['z'].numpy()), 'z_p999': float(np.percentile(np.abs(z - st['z'].numpy()), 99.9) / max(1.0, float(np.abs(st['z'].numpy()).max())))}
            for name, a, b in zip(['main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset'], logits, lref):
                errs[name] = _rel_err(a.cpu().numpy(), b.numpy())
            res[fold] = (errs, len(names))
            record_measurement('neck_up_fold_vs_oracle', 'fold_%s' % fold, errs)
            for name, e in errs.items():
                assert e <= (Z_PEAK_RTOL if name == 'z' else Z_P999_RTOL if name == 'z_p999' else FEAT_RTOL if name.startswith('feat') else LOGIT_RTOL), (fold, name, e)
        finally:
            plan_mod.FOLD_NECK_UP = True
            plan_mod.V2_MIN_TILES = v2_min
    assert res[True][1] == res[False][1] - 3


def _check_device_decode(dev, m, lg, g, K, topk=100):
    """Reference logits in -> (1) the facade (Model.inference + optim_decode_bbox3d, detect.py:61-74) and (2) the
    fused device path (decode2d -> decode3d_slots, no host hop) against the reference's detections (bit-exact),
    its kept 3D boxes (identical set, <= 1e-4) and the raw optimiser state of every detection."""
    dim_ref = rtm3d_amd.kitti_config().DETECTOR.dim_ref
    B = len(g['det_n'])
    d = m.inference(lg)
    det = m.decode2d(lg)
    Kd = torch.as_tensor(np.tile(K, (B, 1)), device=dev)
    boxes = rtm3d_amd.model_utils.decode3d_slots(det, Kd, dim_ref, [0, -0.5, 20])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(det.n.cpu().numpy(), g['det_n'])
    x, fun, st = boxes.x.cpu().numpy(), boxes.fun.cpu().numpy(), boxes.status.cpu().numpy()
    n_kept = 0
    for b in range(B):
        nb = int(g['det_n'][b])
        assert (st[b * topk + nb:(b + 1) * topk] == -1).all()
        if nb == 0:
            assert d[0][b] is None
            continue
        for k, r in enumerate(dets_from_golden(g, 'det_', b)):
            np.testing.assert_array_equal(to_np(d[k][b]), r)
        np.testing.assert_array_equal(det.verts[b * topk:b * topk + nb].cpu().numpy(), g['det_verts_%d' % b])
        # (1) drop-in call of detect.py:71-74
        out = rtm3d_amd.model_utils.optim_decode_bbox3d(to_np(d[0][b]), to_np(d[3][b]), K, dim_ref, [0, -0.5, 20])
        assert out.get_field('class') == g['d3_class_%d' % b].tolist()
        np.testing.assert_allclose(out.get_field('location').reshape(-1, 3), g['d3_location_%d' % b], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out.get_field('dimension').reshape(-1, 3), g['d3_dimension_%d' % b], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out.get_field('Ry'), g['d3_Ry_%d' % b], rtol=0, atol=1e-4)
        # (2) device slots: same keep/reject decision for every detection; kept objects within 1e-4 of the reference's
        # optimum; rejected ones (non-cuboid key points, fun >> 0.1, flat valleys) end at the same objective value
        xs, fs = x[b * topk:b * topk + nb], fun[b * topk:b * topk + nb]
        rx, rf = g['d3_raw_x_%d' % b], g['d3_raw_fun_%d' % b]
        kept = rf < 0.1
        np.testing.assert_array_equal(fs < 0.1, kept)
        np.testing.assert_allclose(xs[kept], rx[kept], rtol=0, atol=1e-4)
        np.testing.assert_allclose(fs, rf, rtol=1e-2, atol=1e-6)
        n_kept += int(kept.sum())
    return n_kept


@pytest.mark.parametrize('fname', ['e2e_dla34_small.npz', 'e2e_resnet18_small.npz', 'e2e_resnet34_small.npz'])
def test_pipeline_on_oracle_logits_matches_reference_golden(dev, fname):
    """Stage parity regime (SURVEY H2 i): decode kernels fed the reference's fp32 logits reproduce the reference's
    detections bit-exactly and the solver state of every detection (these natural detections are not cuboid
    projections: the reference rejects all of them, and so must the device)."""
    g = load_golden(fname)
    lg = [torch.from_numpy(g['logits_' + n]).to(dev) for n in ['main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset']]
    _check_device_decode(dev, make_model('DLA-34'), lg, g, g['K'])


@pytest.mark.parametrize('name', sorted(PLANTED_CASES))
def test_planted_boxes_decode2d_to_decode3d_vs_reference(dev, name):
    """Hand-off decode2d -> decode3d on objects the REFERENCE keeps: cuboid projections planted into reference-run
    logits (tests/golden/cases.py); >= 10 kept boxes per image next to rejected natural detections."""
    g = load_golden(name + '.npz')
    th, tk, K, arrs, _ = planted_inputs(name, load_golden(PLANTED_CASES[name][0]))
    np.testing.assert_array_equal(np.concatenate([a.reshape(-1)[:16] for a in arrs]), g['probe'])
    lg = [torch.from_numpy(a).to(dev) for a in arrs]
    n_kept = _check_device_decode(dev, make_model('DLA-34', None, th, tk), lg, g, K, tk)
    assert n_kept >= 10 * len(g['det_n'])


def test_detect3d_pipeline_and_batch_invariance(dev):
    """Fused device pipeline; images are independent: image b of a batch of 4 == the same image run alone.  The two runs
    need not use the same kernels for every layer (tile-count heuristics, split-K for small batches), so fp32 sums may be
    rounded in a different order: logits agree to fp16 round-off, detections away from the threshold agree in (class,
    position) with vertices within 0.1 px; two runs of the SAME batch are bit-identical."""
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5)
    m = make_model(bb, sd)
    x = weights.synth_images(4, 64, 128, seed=5).to(dev)
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (4, 1)), device=dev)
    det, boxes, logits = m.detect3d(x, K)
    torch.cuda.synchronize()
    n = det.n.cpu().numpy()
    logits = [l.clone() for l in logits]
    cls4, sc4, mp4, v4, x4 = (t.clone() for t in (det.cls, det.score, det.mproj, det.verts, boxes.x))
    det_b, boxes_b, logits_b = m.detect3d(x, K)
    for a, c in zip(logits, logits_b):
        assert torch.equal(a, c)
    assert torch.equal(cls4, det_b.cls) and torch.equal(v4, det_b.verts) and torch.equal(x4, boxes_b.x)
    seen = 0
    for b in range(4):
        d1, b1, l1 = m.detect3d(x[b:b + 1], K[b:b + 1])
        for a, c in zip(logits, l1):
            scale = max(1.0, float(a[b:b + 1].abs().max()))
            assert float((a[b:b + 1] - c).abs().max()) <= 5e-3 * scale
        k1 = int(d1.n.item())
        one = {(int(c), tuple(np.floor(mp / 4).astype(int))): v for c, mp, v in
               zip(d1.cls[:k1].cpu().numpy(), d1.mproj[:k1].cpu().numpy(), d1.verts[:k1].cpu().numpy())}
        k = int(n[b])
        for c, s_, mp, v in zip(cls4[b * 100:b * 100 + k].cpu().numpy(), sc4[b * 100:b * 100 + k].cpu().numpy(),
                                mp4[b * 100:b * 100 + k].cpu().numpy(), v4[b * 100:b * 100 + k].cpu().numpy()):
            if s_ < 0.42:
                continue
            key = (int(c), tuple(np.floor(mp / 4).astype(int)))
            assert key in one, (b, key, s_)
            assert np.abs(one[key] - v).max() <= 0.1
            seen += 1
    assert seen > 0
    st = boxes.status.cpu().numpy().reshape(4, 100)
    for b in range(4):
        assert (st[b, :n[b]] >= 0).all() and (st[b, n[b]:] == -1).all()


def test_full_size_round_trip_properties(dev):
    """BASELINE full size (bs=4 of 384x1280): size-independent properties instead of an oracle run:
    NMS idempotence/sortedness of the decode and consistency of boxes with their vertices."""
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0)
    m = make_model(bb, sd)
    x = weights.synth_images(4, 384, 1280, seed=1234).to(dev)
    (clses, scores, mprojs, verts, boxes), logits = m(x)
    assert logits[0].shape == (4, 3, 96, 320) and logits[1].shape == (4, 16, 96, 320)
    for b in range(4):
        if clses[b] is None:
            continue
        s = scores[b].cpu().numpy()
        assert (np.diff(s) <= 0).all() and (s > 0.4).all() and len(s) <= 100       # sorted, thresholded, capped
        v = verts[b].cpu().numpy(); bx = boxes[b].cpu().numpy()
        np.testing.assert_array_equal(bx[:, :2], v.min(1)); np.testing.assert_array_equal(bx[:, 2:], v.max(1))
        # every detection is a strict 3x3 local maximum of its class plane
        hm = torch.sigmoid(logits[0][b]).cpu()
        for c, (mx, my) in zip(clses[b].cpu().numpy(), mprojs[b].cpu().numpy()):
            xx, yy = int(mx // 4), int(my // 4)
            win = hm[c, max(0, yy - 1):yy + 2, max(0, xx - 1):xx + 2]
            assert float(win.max()) == float(hm[c, yy, xx])


def test_errors_are_loud(dev):
    m = make_model('DLA-34')
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))                    # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 65, 64, device=dev))        # not a multiple of 32
    with pytest.raises(RuntimeError):
        m.load_state_dict({'bogus': torch.zeros(1)})
    lib = _lib.load()
    assert lib.rtm3d_decode2d(None, None, None, None, 1, 3, 8, 8, ctypes.c_float(0.4), 100, ctypes.c_float(4.0), None, None, None, None, None, None, None) != 0
    assert b'null' in lib.rtm3d_last_error()


def test_smoke_head_variant_self_consistency(dev):
    """SURVEY 8 a12 / BASELINE configs[4]: the smoke branch's source is not in the reference snapshot, so
    this is a PARITY-UNPINNED regression check of the HIP path against the oracle's restatement of the
    published SMOKE head layout (2 branches: heat map + 8 regression channels, closed-form box)."""
    from oracle import smoke_ref
    bb = 'DLA-34'
    cfg = rtm3d_amd.kitti_config(bb)
    cfg.MODEL.HEAD_VARIANT = 'smoke'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.0, head_variant='smoke')
    assert 'detect_header.regression_header.regression_head.weight' in sd and len(sd) == 321 - 2 * 16
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    m.load_state_dict(sd)
    x = weights.synth_images(2, 96, 160, seed=21)
    K = np.tile(weights.synth_intrinsics(), (2, 1))
    (clses, scores, mprojs, regs), logits = m(x.to(dev))
    lref = smoke_ref.model_forward(x, sd, bb)
    assert len(logits) == 2 and logits[1].shape == (2, 8, 24, 40)
    for a, b in zip(logits, lref):
        assert (a.cpu() - b).abs().max().item() < LOGIT_RTOL * max(1.0, b.abs().max().item())
    # decode on the oracle's logits: key points bit-exact, boxes to fp64 libm accuracy
    det, boxes, _ = None, None, None
    from rtm3d_amd.model_utils import decode_smoke_slots
    d = m.decode2d([l.to(dev) for l in lref])
    bx = decode_smoke_slots(d, lref[1].to(dev), torch.as_tensor(K, device=dev), cfg.DETECTOR.dim_ref, 4.0)
    ref = smoke_ref.decode(lref[0], lref[1], K, cfg.DETECTOR.dim_ref, 0.4, 100, 4.0)
    n = d.n.cpu().numpy()
    for b in range(2):
        if ref[b] is None:
            assert n[b] == 0
            continue
        k = len(ref[b]['cls'])
        assert n[b] == k
        np.testing.assert_array_equal(d.cls[b * 100:b * 100 + k].cpu().numpy(), ref[b]['cls'])
        np.testing.assert_array_equal(d.score[b * 100:b * 100 + k].cpu().numpy(), ref[b]['score'])
        np.testing.assert_array_equal(d.mproj[b * 100:b * 100 + k].cpu().numpy(), ref[b]['xy'])
        np.testing.assert_allclose(bx.x[b * 100:b * 100 + k].cpu().numpy(), ref[b]['x8'], rtol=1e-9, atol=1e-9)
        assert (bx.status[b * 100 + k:(b + 1) * 100] == -1).all()
    # the fused device pipeline uses the same kernels
    det, boxes, _ = m.detect3d(x.to(dev), torch.as_tensor(K, device=dev))
    assert int(det.n.sum()) == int((boxes.status >= 0).sum())


def test_config1_resnet18_bs8_device_decode_equals_host_decode(dev):
    """BASELINE configs[1]: ResNet-18, bs=8, 384x1280, backbone+heads in HIP, decode on the host.
    The host decode (oracle restatement of Model.inference) of the HIP logits and the device decode
    kernel must agree bit for bit (same logits in, integer/fp32 arithmetic in the same order)."""
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-5.0)
    m = make_model(bb, sd)
    x = weights.synth_images(8, 384, 1280, seed=1234).to(dev)
    dets, logits = m(x)
    host = rtm3d_ref.inference([l.cpu() for l in logits], 0.4, 100, 4.0)
    seen = 0
    for b in range(8):
        if host[0][b] is None:
            assert dets[0][b] is None
            continue
        sc = host[1][b].numpy()
        if len(np.unique(sc)) != len(sc):
            continue
        seen += len(sc)
        for k in range(5):
            np.testing.assert_array_equal(to_np(dets[k][b]), to_np(host[k][b]))
    assert seen > 0


def test_config2_dla34_bs32_full_size_properties(dev):
    """BASELINE configs[2] at full size (bs=32, 384x1280): images are independent, so image b of the batch
    must equal the same image run alone, for a sample of b.  The two runs do not use the same kernels for
    every layer (the tile-count heuristic sends the bs=1 head convs to the 128-pixel kernel, whose K loop
    runs tap-major, the bs=32 ones to the halo kernel, which runs chunk-major), so fp32 sums are rounded in
    a different order: logits agree to fp16 round-off, detections away from the score threshold agree in
    (class, position), vertices to 0.25 px.  Two bs=32 runs are bit-identical."""
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0)
    m = make_model(bb, sd)
    x = weights.synth_images(32, 384, 1280, seed=1234).to(dev)
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (32, 1)), device=dev)
    det, boxes, logits = m.detect3d(x, K)
    torch.cuda.synchronize()
    n = det.n.cpu().numpy()
    assert n.sum() > 0 and n.max() <= 100
    logits = [l.clone() for l in logits]
    cls32, verts32, score32, mp32 = det.cls.clone(), det.verts.clone(), det.score.clone(), det.mproj.clone()
    det2, boxes2, logits2 = m.detect3d(x, K)
    for a, c in zip(logits, logits2):
        assert torch.equal(a, c)
    assert torch.equal(cls32, det2.cls) and torch.equal(verts32, det2.verts)
    # images 0 / 13 / 31 of the batch against the ORACLE's fp32 CPU forward of the same images (this replaces
    # a HIP-vs-HIP comparison at 2e-2): logits at LOGIT_RTOL, detections by the margin rule of the reference-golden tests
    xc = x.cpu()
    thr_logit = float(np.log(0.4 / 0.6))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for b in (0, 13, 31):
        dref, lref = rtm3d_ref.model_forward(xc[b:b + 1], sd, bb)
        errs = {}
        for name, a, c in zip(['main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset'], logits, lref):
            errs[name] = _rel_err(a[b:b + 1].cpu().numpy(), c.numpy())
            assert errs[name] <= LOGIT_RTOL, (b, name, errs[name])
        record_measurement('config2_bs32_vs_oracle', 'image_%d' % b, errs)
        tol = HM_RTOL * max(1.0, float(lref[0].abs().max()))
        kb = int(n[b])
        got = {(int(c), int(mx // 4), int(my // 4)): v for c, (mx, my), v in
               zip(cls32[b * 100:b * 100 + kb].cpu().numpy(), mp32[b * 100:b * 100 + kb].cpu().numpy(), verts32[b * 100:b * 100 + kb].cpu().numpy())}
        if dref[0][0] is not None:
            rs = dref[1][0].numpy().astype(np.float64)
            sure = np.abs(np.log(rs / (1.0 - rs)) - thr_logit) > tol
            for c, mp, v, ok in zip(dref[0][0].numpy(), dref[2][0].numpy(), dref[3][0].numpy(), sure):
                if ok:
                    key = (int(c), int(mp[0] // 4), int(mp[1] // 4))
                    assert key in got, (b, key)
                    assert np.abs(got[key] - v).max() < VERT_TOL_PX
        d1, b1, l1 = m.detect3d(x[b:b + 1], K[b:b + 1])
        k = int(n[b])
        sc = score32[b * 100:b * 100 + k].cpu().numpy()
        key32 = {(int(c), tuple(np.floor(mp / 4).astype(int))): v for c, mp, v, s_ in
                 zip(cls32[b * 100:b * 100 + k].cpu().numpy(), mp32[b * 100:b * 100 + k].cpu().numpy(),
                     verts32[b * 100:b * 100 + k].cpu().numpy(), sc) if s_ >= 0.45}
        k1 = int(d1.n.item())
        key1 = {(int(c), tuple(np.floor(mp / 4).astype(int))): v for c, mp, v in
                zip(d1.cls[:k1].cpu().numpy(), d1.mproj[:k1].cpu().numpy(), d1.verts[:k1].cpu().numpy())}
        assert len(key32) > 0
        for kk, v in key32.items():
            assert kk in key1, (b, kk)
            assert np.abs(key1[kk] - v).max() <= 0.25


@pytest.mark.parametrize('B', [3, 1])
def test_two_stream_pipeline_equals_serial_path(dev, B):
    """The bench path (forward ∥ decode3d on two streams, double-buffered slots) returns exactly the records
    of the serial path, for a sequence of different batches (exercises slot reuse and event ordering).  B = 1 takes the
    small-batch configuration: hipGraph replay, three slots, one decode stream per slot."""
    from rtm3d_amd.pipeline import Detect3DPipeline
    from rtm3d_amd import distributed as rdist
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5)
    m = make_model(bb, sd)
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), device=dev)
    xs = [weights.synth_images(B, 64, 128, seed=100 + 7 * i).to(dev) for i in range(7)]
    pipe = Detect3DPipeline(m, B, dev, gather=True)
    assert (pipe.depth, len(pipe.sides)) == ((3, 3) if B == 1 else (2, 1))
    ids = [pipe.submit(x, K) for x in xs[:2]]
    got = {ids[0]: pipe.results(ids[0]).clone()}            # read slot 0 before it is reused
    for x in xs[2:]:
        i = pipe.submit(x, K)
        got[i - 1] = pipe.results(i - 1).clone()
        ids.append(i)
    got[ids[-1]] = pipe.results(ids[-1]).clone()
    pipe.drain()
    for i, x in enumerate(xs):
        det, boxes, _ = m.detect3d(x, K)
        ref = pack_records_reference(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, 100, boxes)
        torch.cuda.synchronize()
        assert torch.equal(got[i], ref), i
    recs = rdist.unpack_records(got[0])
    assert len(recs) == B and all(r is None or r['verts'].shape[1:] == (8, 2) for r in recs)


def test_two_stream_pipeline_with_saturated_topk(dev):
    """Maximum load of the decode: a dense heat map fills all 100 slots of every image (the regime of a trained checkpoint on
    a crowded scene, bench.py --heat-bias 2): 400 objects per batch through decode2d -> decode3d_slots -> pack, pipelined
    records equal to the serial path, every slot solved."""
    from rtm3d_amd.pipeline import Detect3DPipeline
    bb = 'RESNET-18'
    m = make_model(bb, weights.synth_state_dict(bb, 1, 'trained', heat_bias=2.0))
    B = 4
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), device=dev)
    xs = [weights.synth_images(B, 64, 128, seed=300 + i).to(dev) for i in range(3)]
    pipe = Detect3DPipeline(m, B, dev, gather=False)
    got = []
    for x in xs:
        got.append(pipe.results(pipe.submit(x, K)).clone())
    pipe.drain()
    for x, g_ in zip(xs, got):
        det, boxes, _ = m.detect3d(x, K)
        ref = pack_records_reference(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, 100, boxes)
        torch.cuda.synchronize()
        assert det.n.cpu().tolist() == [100] * B
        assert (boxes.status.cpu().numpy() >= 0).all()
        assert torch.equal(g_, ref)


def test_model_detect_gives_the_lists_of_forward(dev):
    """Model.detect(x) = what detect.py:56 takes (`model(imgs)[0]`), through the peaks-only regression heads: same per-image
    lists (None for an empty image), classes / scores identical, key points / vertices / boxes to fp16 round-off."""
    bb = 'RESNET-18'
    m = make_model(bb, weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    x = weights.synth_images(3, 64, 128, seed=12).to(dev)
    x[1] = 0                                        # an image without detections? (decided by the heat map; either way both agree)
    ref = m(x)[0]
    got = m.detect(x)
    assert len(got) == 5
    for b in range(3):
        if ref[0][b] is None:
            assert all(g[b] is None for g in got)
            continue
        # (the heat-map-only plan may pick other tile shapes for a small launch: scores to fp32 summation order, not bit for bit)
        assert torch.equal(got[0][b], ref[0][b]) and float((got[1][b] - ref[1][b]).abs().max()) <= 2e-3
        for k in (2, 3, 4):
            assert float((got[k][b] - ref[k][b]).abs().max()) <= 0.05
    assert any(r is not None for r in ref[0])


def test_sparse_heads_with_no_detections(dev):
    """Empty images through the peaks-only path: no peak above the threshold anywhere -> every slot is empty, the patch plan runs
    on whatever its buffers hold and nobody reads its results: counts 0, lists None, records all zero, no 3D solve."""
    from rtm3d_amd import distributed as rdist
    bb = 'RESNET-18'
    m = make_model(bb, weights.synth_state_dict(bb, 1, 'trained', heat_bias=-30.0))
    x = weights.synth_images(2, 64, 128, seed=3).to(dev)
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (2, 1)), device=dev)
    m.detect3d(weights.synth_images(2, 64, 128, seed=4).to(dev), K, sparse_heads=True)        # leave something in the patch buffers
    det, boxes, _ = m.detect3d(x, K, sparse_heads=True)
    rec = rdist.pack_records(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, det.topk, boxes)
    torch.cuda.synchronize()
    assert det.n.cpu().tolist() == [0, 0] and (boxes.status.cpu().numpy() == -1).all() and float(rec.abs().sum()) == 0.0
    assert all(v == [None, None] for v in m.detect(x))


def test_two_stream_pipeline_with_sparse_heads_equals_serial_sparse_path(dev):
    """Detect3DPipeline(sparse_heads=True): forward (heat map only) -> peaks -> patch plan -> finish on the main stream, 3D decode
    and packing on the side stream, over several pipelined steps with different inputs: records equal to the serial
    detect3d(sparse_heads=True) + pack of the same batch, bit for bit (the patch plan's buffers are per plan, the slots per
    pipeline: nothing of step i may leak into step i + 1)."""
    from rtm3d_amd import distributed as rdist
    from rtm3d_amd.pipeline import Detect3DPipeline
    bb = 'RESNET-18'
    m = make_model(bb, weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    B = 3
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), device=dev)
    xs = [weights.synth_images(B, 64, 128, seed=500 + 11 * i).to(dev) for i in range(4)]
    pipe = Detect3DPipeline(m, B, dev, gather=False, sparse_heads=True)
    got = {}
    for i, x in enumerate(xs):
        k = pipe.submit(x, K)
        if k >= 1:
            got[k - 1] = pipe.results(k - 1).clone()
    got[len(xs) - 1] = pipe.results(len(xs) - 1).clone()
    pipe.drain()
    total = 0
    for i, x in enumerate(xs):
        det, boxes, _ = m.detect3d(x, K, sparse_heads=True)
        ref = pack_records_reference(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, 100, boxes)
        torch.cuda.synchronize()
        assert torch.equal(got[i], ref), i
        total += int(det.n.sum())
    assert total >= 10


def test_pack_records_hip_equals_reference(dev):
    """rtm3d_pack_records (one HIP launch) against the plain-torch definition of the record: every field incl. the fp64
    atan2, the kept flag at the fun < 0.1 edge and the zeroing of empty slots."""
    from rtm3d_amd import distributed as rdist
    from rtm3d_amd.model_utils import Boxes3D
    g = torch.Generator().manual_seed(5)
    B, topk = 5, 100
    n = torch.tensor([0, 100, 37, 1, 64], dtype=torch.int32)
    cls = torch.randint(0, 3, (B * topk,), generator=g)
    score = torch.rand(B * topk, generator=g); mproj = torch.rand(B * topk, 2, generator=g) * 1280
    verts = (torch.rand(B * topk, 8, 2, generator=g) - 0.3) * 1500; bbox = torch.rand(B * topk, 4, generator=g) * 1280
    boxes = Boxes3D(B * topk, dev)
    boxes.x.copy_((torch.randn(B * topk, 8, generator=g, dtype=torch.float64) * 10).to(dev))
    fun = torch.rand(B * topk, generator=g, dtype=torch.float64) * 0.2
    fun[100] = 0.1; fun[101] = float(np.nextafter(0.1, 0.0)); fun[102] = float('inf')
    boxes.fun.copy_(fun.to(dev))
    boxes.status.copy_(torch.randint(-1, 3, (B * topk,), generator=g).to(torch.int32).to(dev))
    t = [a.to(dev) for a in (n, cls, score, mproj, verts, bbox)]
    for bx in (boxes, None):
        got = rdist.pack_records(*t, topk, bx)
        ref = pack_records_reference(*t, topk, bx)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        assert bx is None or int((got[..., 31] == 2).sum()) > 0
    with pytest.raises(RuntimeError):
        rdist.pack_records(n, cls, score, mproj, verts, bbox, topk)          # CPU tensors: no CPU path


def test_nonfinite_logits_through_decode2d_decode3d_pack(dev):
    """NaN / Inf logits and key points through decode2d -> decode3d -> pack, against the oracle on the
    same arrays: a NaN heat-map cell is no peak on either side (NaN > thresh is False), a +Inf cell is a peak with score 1;
    NaN / Inf regression values at a peak give non-finite vertices, for which the solver stops at once with status 3, x = x0,
    fun = NaN, nit = 0 (what SciPy returns), the record carries flag 1 (2D only) and no kept box; finite objects of the same
    image are solved as usual.  The launch terminates (bounded work per object: 0 iterations)."""
    from rtm3d_amd import distributed as rdist
    from rtm3d_amd.model_utils import decode3d_slots
    g = load_golden('planted_small.npz')
    th, tk, K, arrs, _ = planted_inputs('planted_small', load_golden(PLANTED_CASES['planted_small'][0]))
    arrs = [a[:1].copy() for a in arrs]
    d0 = rtm3d_ref.inference([torch.from_numpy(a) for a in arrs], th, tk, 4.0)
    n0 = len(d0[0][0])
    cells = (d0[2][0].numpy() // 4).astype(int)                     # (n, 2) = x, y of every peak
    # poison: NaN vertex offsets at peak 0, +Inf at peak 1, NaN main offset at peak 2; heat map: one NaN cell that was a
    # peak (peak 3 disappears), one +Inf cell somewhere else (a new peak with score 1)
    x0_, y0_ = cells[0]; arrs[1][0, 3, y0_, x0_] = np.nan
    x1_, y1_ = cells[1]; arrs[1][0, 0, y1_, x1_] = np.inf
    x2_, y2_ = cells[2]; arrs[2][0, 1, y2_, x2_] = np.nan
    x3_, y3_ = cells[3]; arrs[0][0, int(d0[0][0][3]), y3_, x3_] = np.nan
    arrs[0][0, 1, 5, 7] = np.inf
    d_ref = rtm3d_ref.inference([torch.from_numpy(a) for a in arrs], th, tk, 4.0)
    m = make_model('DLA-34', None, th, tk)
    det = m.decode2d([torch.from_numpy(a).to(dev) for a in arrs])
    Kd = torch.as_tensor(K.reshape(1, 9), device=dev)
    bx = decode3d_slots(det, Kd, m.config.DETECTOR.dim_ref, [0, -0.5, 20])
    rec = rdist.pack_records(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, tk, bx)
    torch.cuda.synchronize()
    n = int(det.n.item())
    assert n == len(d_ref[0][0]) and n in (n0, n0 + 1, n0 - 1)
    # detections: same classes and cells; finite fields bit-exact, non-finite ones non-finite in the same places
    assert torch.equal(det.cls[:n].cpu(), d_ref[0][0])
    for got, ref in ((det.score[:n], d_ref[1][0]), (det.mproj[:n], d_ref[2][0]), (det.verts[:n], d_ref[3][0])):
        got, ref = got.cpu().numpy(), ref.numpy()
        np.testing.assert_array_equal(np.isfinite(got), np.isfinite(ref))
        np.testing.assert_array_equal(got[np.isfinite(ref)], ref[np.isfinite(ref)])
    assert float(det.score[:n].max()) == 1.0                        # the +Inf cell
    verts = det.verts[:n].cpu().numpy().reshape(n, 16)
    bad = ~np.isfinite(verts).all(1)
    assert 2 <= bad.sum() <= 3
    st, fun, nit, xs = (t[:n].cpu().numpy() for t in (bx.status, bx.fun, bx.nit, bx.x))
    assert (st[bad] == 3).all() and (nit[bad] == 0).all() and not np.isfinite(fun[bad]).any()
    _, raw = decode3d_ref.optim_decode_bbox3d(d_ref[0][0].numpy(), d_ref[3][0].numpy(), K, m.config.DETECTOR.dim_ref, [0, -0.5, 20],
                                              return_raw=True)
    assert (raw['nit'][bad] == 0).all() and not raw['kept'][bad].any()
    np.testing.assert_array_equal(xs[bad], raw['x'][bad])
    np.testing.assert_array_equal(fun[~bad] < 0.1, raw['kept'][~bad])
    good_kept = (~bad) & raw['kept']
    assert good_kept.sum() >= 5
    np.testing.assert_allclose(xs[good_kept], raw['x'][good_kept], rtol=0, atol=1e-4)
    r = rec[0, :n].cpu().numpy()
    assert (r[bad, 31] == 1).all() and (r[good_kept, 31] == 2).all() and (rec[0, n:] == 0).all()


@pytest.mark.parametrize('bb,B,H,W,hb', [('RESNET-18', 3, 128, 256, -3.0), ('DLA-34', 2, 128, 256, -3.0), ('DLA-34', 1, 64, 128, 1.5),
                                        ('DLA-34', 8, 384, 1280, -5.0)])
def test_sparse_heads_equal_dense_heads_at_the_peaks(dev, bb, B, H, W, hb):
    """Peaks-only regression heads (detect3d(sparse_heads=True), csrc/sparse_heads.hip) against the dense path of the same
    model on the same images: the heat-map branch is the same kernels on the same operands, so detections (count, class,
    score, cell) are IDENTICAL; the regression logits at the peaks come from the patch plan (other tile shapes, same fp16
    operands, fp32 accumulation in another order), so sub-pixel key points / vertices agree to fp16 round-off of two
    256-channel layers: bar 0.05 px (the dense path itself is within 0.038 px of the fp32 reference), kept 3D boxes follow.
    Border and corner peaks are forced by the heat bias of the third case (top-k saturates: 100 peaks on a 16 x 32 map)."""
    from rtm3d_amd.model_utils import decode3d_slots
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=hb)
    m = make_model(bb, sd)
    x = weights.synth_images(B, H, W, seed=91).to(dev)
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), dtype=torch.float64, device=dev)
    _, _, lg_d = m.detect3d(x, K)
    det_s, box_s, lg_s = m.detect3d(x, K, sparse_heads=True)
    torch.cuda.synchronize()
    if hb > 0:
        # peaks planted on every border and in every corner of the map (on top of the network's own): the gather's zero fill
        # beyond the padded tensor and both zero-padding masks decide these
        hm = lg_s[0].clone()
        Hm, Wm = H // 4, W // 4
        cells = [(0, 0), (0, Wm - 1), (Hm - 1, 0), (Hm - 1, Wm - 1)] + [(0, c) for c in range(3, Wm - 2, 4)] + \
                [(Hm - 1, c) for c in range(2, Wm - 2, 4)] + [(r, 0) for r in range(2, Hm - 2, 3)] + [(r, Wm - 1) for r in range(3, Hm - 2, 3)] + \
                [(1, 1), (1, Wm - 2), (Hm - 2, 1), (Hm - 2, Wm - 2)]
        for k_, (r, c) in enumerate(cells):
            hm[0, k_ % 3, max(r - 1, 0):r + 2, max(c - 1, 0):c + 2] = -9.0
            hm[0, k_ % 3, r, c] = 6.0 + 0.01 * k_
        det_s = m.decode2d_sparse((hm,), from_forward=lg_s)             # the plan still holds z of the forward above
        with pytest.raises(RuntimeError, match='does not hold the fused map'):
            m.decode2d_sparse((hm,))                                     # an unrelated tensor: refused
        lg_s = (hm,)
        box_s = decode3d_slots(det_s, K, m.config.DETECTOR.dim_ref, [0, -0.5, 20])
        torch.cuda.synchronize()
        lg_d = (hm,) + tuple(lg_d[1:])
    # the heat map: the same layers on the same operands; a plan with one dense branch may pick another tile shape / split-K
    # for a small launch, so equal to fp32 summation order (and fp16 re-rounding of the two hidden maps), not bit for bit
    assert len(lg_s) == 1 and float((lg_s[0] - lg_d[0]).abs().max()) <= 2e-2 * max(1.0, float(lg_d[0].abs().max()))
    # reference for the regression part: the dense decode fed THIS heat map and the dense regression maps
    det_d = m.decode2d((lg_s[0], lg_d[1], lg_d[2]))
    box_d = decode3d_slots(det_d, K, m.config.DETECTOR.dim_ref, [0, -0.5, 20])
    torch.cuda.synchronize()
    n = det_d.n.cpu().numpy()
    assert np.array_equal(det_s.n.cpu().numpy(), n) and n.sum() >= 3 * B
    tk = det_d.topk
    worst_v, worst_m, on_border = 0.0, 0.0, 0
    for b in range(B):
        sl = slice(b * tk, b * tk + int(n[b]))
        assert torch.equal(det_s.cls[sl], det_d.cls[sl]) and torch.equal(det_s.score[sl], det_d.score[sl])
        md, ms = det_d.mproj[sl].cpu().numpy(), det_s.mproj[sl].cpu().numpy()
        assert np.array_equal(np.floor(md / 4), np.floor(ms / 4))               # same cell
        cell = np.floor(md / 4)
        on_border += int(((cell[:, 0] == 0) | (cell[:, 1] == 0) | (cell[:, 0] == W // 4 - 1) | (cell[:, 1] == H // 4 - 1)).sum())
        worst_m = max(worst_m, float(np.abs(md - ms).max()))
        worst_v = max(worst_v, float(np.abs(det_d.verts[sl].cpu().numpy() - det_s.verts[sl].cpu().numpy()).max()))
        bd, bs = det_d.bbox[sl].cpu().numpy(), det_s.bbox[sl].cpu().numpy()
        assert np.abs(bd - bs).max() <= 0.05
    record_measurement('sparse_heads_vs_dense', '%s_%dx%dx%d' % (bb, B, H, W), {'vert_linf_px': worst_v, 'mproj_linf_px': worst_m,
                                                                                'detections': int(n.sum()), 'border_peaks': on_border})
    assert worst_v <= 0.05 and worst_m <= 0.02, (worst_v, worst_m)
    if hb > 0:
        assert on_border >= 20                                                   # the masks and the zero fill were exercised
    # 3D decode on both: same keep decisions wherever the objective is not within 2 % of the acceptance threshold
    fd, fs = box_d.fun.cpu().numpy(), box_s.fun.cpu().numpy()
    for b in range(B):
        sl = slice(b * tk, b * tk + int(n[b]))
        clear = np.abs(fd[sl] - 0.1) > 0.002
        assert np.array_equal((fd[sl] < 0.1)[clear], (fs[sl] < 0.1)[clear])


def test_pipeline_rejects_wrong_batch(dev):
    """a shard larger than the preallocated slots would be an out-of-bounds device write, a smaller one
    would leave stale detections in the unused rows - both must raise before anything is launched."""
    from rtm3d_amd.pipeline import Detect3DPipeline
    from rtm3d_amd.model import Detections
    bb = 'RESNET-18'
    m = make_model(bb, weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    pipe = Detect3DPipeline(m, 3, dev, gather=False)
    K3 = torch.as_tensor(np.tile(weights.synth_intrinsics(), (3, 1)), device=dev)
    for nb in (2, 4):
        with pytest.raises(ValueError):
            pipe.submit(torch.zeros(nb, 3, 64, 128, device=dev), K3)
    with pytest.raises(ValueError):
        pipe.submit(torch.zeros(3, 3, 64, 128, device=dev), K3[:2])
    lg = [torch.zeros(2, c, 16, 32, device=dev) for c in (3, 16, 2, 2)]
    with pytest.raises(ValueError):
        m.decode2d(lg, out=Detections(3, 100, dev))
    with pytest.raises(ValueError):
        m.decode2d(lg, out=Detections(2, 50, dev))
    assert pipe.submit(torch.zeros(3, 3, 64, 128, device=dev), K3) == 0
    pipe.drain()


def test_pipeline_under_nccl_world1(dev):
    """Multi-GPU readiness on a one-GPU box: Detect3DPipeline with the record all-gather forced through RCCL (backend
    "nccl", world size 1) on the side stream, event-ordered against the main stream, equals the serial path."""
    import subprocess, sys, os
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    port = str(29600 + os.getpid() % 300)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'nccl_world1_worker.py'), port], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'nccl world-1 pipeline ok' in r.stdout


def test_bench_spawn_path_nccl_world1(dev):
    """`bench.py --gpus 1 --force-launch`: the launcher bench.py uses for --gpus N > 1 (fresh rank processes under
    torch.distributed.run, started before the parent touches the GPU), at the one rank this box has: the rank creates the
    RCCL process group, runs the pipelined steps with the preallocated all-gather and reports n_gpus / ranks_seen."""
    import subprocess, sys, os, json
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-launch', '--steps', '3', '--warmup', '1',
                        '--batch', '4', '--height', '128', '--width', '256', '--backbone', 'RESNET-18', '--no-cpu-baseline', '--no-parity'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['multi_gpu']['ranks_seen'] == 1 and out['multi_gpu']['gathered_shape'] == [4, 100, 32]
    assert out['multi_gpu']['allgather_us_last_step'] > 0 and out['value'] > 0


def test_bench_line_parity_of_the_default_solver(dev):
    """The default `python bench.py` line (DLA-34 bs=32 384x1280; few steps, no CPU baseline here) runs the PUBLISHED solver form in
    its timed step and its stage parity - the device decode kernels on the oracle's logits with 16 cuboids planted per image -
    meets north_star on EVERY box the reference keeps: identical indices and keep decisions, box L-inf <= 1e-4."""
    import subprocess, sys, os, json
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-sparse-probe'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    st = out['parity']['stage']
    assert out['config']['solver_form'] == 'published' and st['solver_form'] == 'published'
    assert st['index_mismatches'] == 0 and st['keep_decision_mismatches'] == 0 and st['kept_ref'] >= 100, st
    assert st['box_linf'] <= 1e-4 and st['boxes']['box_linf']['frac_le_0.0001'] == 1.0, st


def test_bench_two_ranks_rehearsed_on_one_gpu(dev):
    """`bench.py --gpus 2 --rehearse-one-gpu`: the launcher starts TWO rank processes that both run the whole hot path on GPU 0 on
    their own shards of the global batch (images rank * B ...) and exchange their records per step from the pipeline's side
    stream - over gloo through host memory, because RCCL refuses two ranks on one device.  Everything of the N > 1 path but the
    transport runs: preallocated (2B, 100, 32) gather buffers, ordering by global image index, both ranks' blocks intact in
    rank 0's gathered batch (checksums gathered separately), per-rank timing; the line says INVALID (no throughput claim)."""
    import subprocess, sys, os, json
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rehearse-one-gpu', '--steps', '3', '--warmup', '1',
                        '--batch', '4', '--height', '128', '--width', '256', '--backbone', 'RESNET-18', '--no-cpu-baseline', '--no-parity',
                        '--no-sparse-probe', '--shared-weight-cache'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    mg = out['multi_gpu']
    assert out['n_gpus'] == 2 and mg['ranks_seen'] == 2 and mg['gathered_shape'] == [8, 100, 32] and len(mg['per_rank_ms_per_step']) == 2
    assert out['config']['global_batch'] == 8 and 'INVALID' in out
    # round 5 (--shared-weight-cache): rank 0 folds + packs the weights once and writes the cache file, rank 1 builds its plan
    # from it (no misses); and the N > 1 line keeps rank 0's roofline block
    wc = {n['rank']: n for n in mg['shared_weight_cache']['per_rank']}
    assert wc[0]['misses'] > 0 and wc[1]['misses'] == 0 and wc[1]['hits'] > 0, wc
    assert out['roofline'] and out['roofline']['launch_ms'] > 0 and out['roofline']['per_stage']['heads']['ms'] > 0


def test_forward_marks_stage_wall_times(dev):
    """The stage marks of one real eager replay (rtm3d_forward_marks: what bench.py reports as per_stage[*].wall_ms): positive,
    the neck's wall time within noise of the sum of its ops' own times, logits identical to an ordinary forward; bad mark lists
    are refused."""
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.0)
    m = make_model(bb, sd)
    m.use_graph = False
    B, H, W = 32, 384, 1280
    x = weights.synth_images(B, H, W, seed=77).to(dev)
    plan = m._plan_for(B, H, W, dev)
    names = plan.op_names
    ref = [t.clone() for t in m.forward_logits(x)]
    torch.cuda.synchronize()
    outs = [torch.empty(B, c, H // 4, W // 4, dtype=torch.float32, device=dev) for c in m._head_channels]
    st = torch.cuda.current_stream(dev).cuda_stream
    ptrs = [o.data_ptr() for o in outs]
    first = lambda pre: next(i for i, n in enumerate(names) if n.startswith(pre))
    marks = [first('backbone'), first(('kfpn', 'fusion')), first('heads')]
    wall = [plan.forward_marks(st, x.data_ptr(), ptrs, marks) for _ in range(4)][-1]
    info = plan.forward_timed(st, x.data_ptr(), ptrs)
    neck_sum = sum(i['ms'] for i in info if i['name'].startswith(('kfpn', 'fusion')))
    assert all(v > 0 for v in wall) and wall[1] < neck_sum * 1.15 + 0.05, (wall, neck_sum)
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)
    lib = _lib.load()
    assert lib.rtm3d_forward_marks(plan.ctx, ctypes.c_void_p(st), ctypes.c_void_p(x.data_ptr()), (ctypes.c_void_p * 4)(*ptrs), 2,
                                   (ctypes.c_int * 2)(5, 5), (ctypes.c_float * 2)()) != 0      # not ascending
    assert lib.rtm3d_forward_marks(plan.ctx, ctypes.c_void_p(st), ctypes.c_void_p(x.data_ptr()), (ctypes.c_void_p * 4)(*ptrs), 1,
                                   (ctypes.c_int * 1)(len(names)), (ctypes.c_float * 1)()) != 0


def test_graph_replay_is_bit_identical_and_plan_cache_is_bounded(dev):
    """hipGraph replay of the plan (small batches) against the eager replay: same logits bit for bit, over several
    replays, fresh input/output buffers (new graph keys) and a second shape; the plan cache keeps at most MAX_PLANS."""
    from rtm3d_amd import model as model_mod
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.0)
    mg, me = make_model(bb, sd), make_model(bb, sd)
    mg.use_graph, me.use_graph = True, False
    xs = [weights.synth_images(2, 64, 128, seed=40 + i).to(dev) for i in range(3)]
    keep = []
    for rep in range(3):
        for x in xs:
            a, b = mg.forward_logits(x), me.forward_logits(x)
            torch.cuda.synchronize()
            for u, v in zip(a, b):
                assert torch.equal(u, v)
            keep.append(a)                       # holding the outputs forces new buffers, i.e. new graph keys
    d_g, d_e = mg(xs[0])[0], me(xs[0])[0]
    for u, v in zip(d_g, d_e):
        assert (u[0] is None and v[0] is None) or torch.equal(u[0], v[0])
    for k, hw in enumerate([(64, 160), (96, 128), (32, 64), (64, 64)]):
        mg.forward_logits(torch.zeros(1, 3, hw[0], hw[1], device=dev))
        assert len(mg._plans) <= model_mod.MAX_PLANS
    a = mg.forward_logits(xs[1]); b = me.forward_logits(xs[1])      # the first shape was evicted and is rebuilt
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('bb,B,H,W', [('RESNET-18', 3, 128, 256), ('DLA-34', 1, 64, 128)])
def test_graph_relaunch_orders_of_mixed_structure(dev, bb, B, H, W):
    """One context, up to eight live graph execs of two structures (fp32 input: with the NHWC4 conversion node; preloaded uint8
    input: without it) and several input / output addresses, captured and re-launched in the orders that failed in round 3
    while the counter reset was a hipMemsetAsync NODE (n, n, n-1 nodes -> re-launch the first: its conversion kernel had no
    effect; n, n-1, n -> re-launch the second: memory access fault in its first convolution) and in a seeded random order;
    every result equals the eager replay bit for bit (tools/gpu_graph_stress.py)."""
    from tools.gpu_graph_stress import run
    assert run(lambda s: None, bb, B, H, W, rounds=30) == 0


def test_graph_with_memset_node_is_refused(dev):
    """ADVICE r03 / the rule in include/rtm3d_hip.h (rtm3d_ctx_set_graph): replay graphs hold kernel nodes only.  With the test
    hook that puts a hipMemsetAsync in front of every replay, the capture contains a memset node: the runtime must drop the
    graph, leave graph mode (graph_stats: enabled = False) and serve the call - and every later one - by the eager replay, with
    results equal to a context that never used graphs; without the hook the same plan captures normally (captures >= 1)."""
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.0)
    mg, me = make_model(bb, sd), make_model(bb, sd)
    mg.use_graph, me.use_graph = True, False
    x = weights.synth_images(1, 64, 128, seed=3).to(dev)
    ref = [l.clone() for l in me.forward_logits(x)]
    plan = mg._plan_for(1, 64, 128, dev)
    lib = _lib.load()
    _lib.check(lib.rtm3d_ctx_debug_memset_in_replay(plan.ctx, 1), 'debug_memset_in_replay')
    for _ in range(3):
        got = mg.forward_logits(x)
        torch.cuda.synchronize()
        for u, v in zip(got, ref):
            assert torch.equal(u, v)
    cap, hits, enabled = plan.graph_stats()
    assert enabled is False and hits == 0, (cap, hits, enabled)
    assert b'non-kernel graph node' in lib.rtm3d_last_error()
    # the same plan without the hook: graphs work (a fresh context: this one has left graph mode for good)
    m2 = make_model(bb, sd)
    m2.use_graph = True
    outs = m2.forward_logits(x, out='reuse')
    outs = m2.forward_logits(x, out='reuse')
    torch.cuda.synchronize()
    cap2, hits2, enabled2 = m2._plan_for(1, 64, 128, dev).graph_stats()
    assert enabled2 is True and cap2 >= 1 and hits2 >= 1, (cap2, hits2, enabled2)
    for u, v in zip(outs, ref):
        assert torch.equal(u, v)


def test_forward_logits_out_keeps_one_graph(dev):
    """a bs=1 loop that HOLDS its outputs gets fresh logit tensors on every call, i.e. a new graph key and
    a capture per call until the context gives up on graphs after 32; with out='reuse' (or explicit out= tensors) the same
    loop replays ONE captured graph.  Logits are bit-identical on both routes."""
    sd = weights.synth_state_dict('RESNET-18', 1, 'trained', heat_bias=-3.0)
    x = weights.synth_images(1, 64, 128, seed=77).to(dev)
    m = make_model('RESNET-18', sd)
    m.use_graph = True
    ref = [t.clone() for t in m.forward_logits(x)]
    held = []
    for _ in range(40):
        held.append(m.forward_logits(x))                  # the trap: 40 live output sets = 40 distinct keys
    torch.cuda.synchronize()
    plan = m._plan_for(1, 64, 128, dev)
    cap, hits, on = plan.graph_stats()
    assert cap > 32 and not on                            # gave up: eager replay from here on
    for a, b in zip(held[-1], ref):
        assert torch.equal(a, b)
    m2 = make_model('RESNET-18', sd)
    m2.use_graph = True
    held = []
    for _ in range(40):
        lg = m2.forward_logits(x, out='reuse')
        held.append(lg[0].data_ptr())
    mine = [torch.empty_like(t) for t in ref]
    for _ in range(5):
        lg2 = m2.forward_logits(x, out=mine)
    torch.cuda.synchronize()
    cap, hits, on = m2._plan_for(1, 64, 128, dev).graph_stats()
    assert len(set(held)) == 1 and cap == 2 and hits == 43 and on
    for a, b, c in zip(lg, lg2, ref):
        assert torch.equal(a, c) and torch.equal(b, c)
    with pytest.raises(ValueError):
        m2.forward_logits(x, out=mine[:2])
    with pytest.raises(ValueError):
        m2.forward_logits(x, out=[t.double() for t in mine])


def test_class_count_follows_dataset_objs(dev):
    """the heat-map head has len(cfg.DATASET.OBJs) channels (models/nets/header.py:11); a class list of
    5 runs end to end, and a dim_ref with fewer rows than classes raises like the reference's dim_ref[cls] would."""
    bb = 'RESNET-18'
    cfg = rtm3d_amd.kitti_config(bb)
    cfg.DATASET.OBJs = ['Car', 'Pedestrian', 'Cyclist', 'Van', 'Truck']
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    sd = weights.synth_state_dict(bb, 4, 'trained', heat_bias=-2.5, num_classes=5)
    m.load_state_dict(sd)
    x = weights.synth_images(2, 64, 128, seed=8)
    dets, logits = m(x.to(dev))
    assert logits[0].shape == (2, 5, 16, 32)
    dref, lref = rtm3d_ref.model_forward(x, sd, bb)
    for a, b in zip(logits, lref):
        assert _rel_err(a.cpu().numpy(), b.numpy()) <= LOGIT_RTOL
    d = m.inference([l.to(dev) for l in lref])                 # decode of the oracle's logits: bit-exact, classes 0..4
    seen = set()
    for b in range(2):
        if dref[0][b] is None:
            assert d[0][b] is None
            continue
        assert torch.equal(d[0][b].cpu(), dref[0][b]) and torch.equal(d[3][b].cpu(), dref[3][b])
        seen |= set(dref[0][b].tolist())
    assert max(seen) >= 3
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (2, 1)), device=dev)
    with pytest.raises(IndexError):
        m.detect3d(x.to(dev), K)                               # kitti dim_ref has 3 rows
    with pytest.raises(IndexError):
        rtm3d_amd.model_utils.optim_decode_bbox3d(np.array([4]), np.zeros((1, 8, 2), np.float32), weights.synth_intrinsics(),
                                                  cfg.DETECTOR.dim_ref, [0, -0.5, 20])
    dim5 = list(cfg.DETECTOR.dim_ref) + [[2.0, 1.9, 5.0], [3.2, 2.5, 9.0]]
    det, boxes, _ = m.detect3d(x.to(dev), K, dim_ref=dim5)
    assert int(det.n.sum()) == int((boxes.status >= 0).sum())
    # the peaks-only regression heads with five classes and another top-k (the patch plan has B * topk slots)
    cfg.DETECTOR.TOPK_CANDIDATES = 37
    det_d, _, _ = m.detect3d(x.to(dev), K, dim_ref=dim5)
    det_s, box_s, lg = m.detect3d(x.to(dev), K, dim_ref=dim5, sparse_heads=True)
    torch.cuda.synchronize()
    assert lg[0].shape == (2, 5, 16, 32) and det_s.topk == 37 and torch.equal(det_s.n, det_d.n) and int(det_s.n.max()) <= 37
    for b in range(2):
        sl = slice(b * 37, b * 37 + int(det_d.n[b]))
        assert torch.equal(det_s.cls[sl], det_d.cls[sl])
        assert float((det_s.verts[sl] - det_d.verts[sl]).abs().max()) <= 0.05


def test_project_boxes_device_vs_reference_vectors(dev):
    """n3 on the device: rtm3d_project_boxes against the vectors produced by running the reference's calc_proj_corners
    (tests/golden/project_cases.npz, incl. yaw values inside the 1e-3 snapping window) and against the host form on the
    solver's own outputs; fp64, 1e-9 relative (the device's sin/cos/atan2 are not numpy's)."""
    from rtm3d_amd import kitti_results as kr
    from rtm3d_amd.model_utils import Boxes3D
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'project_cases.npz'))
    n = len(g['Ry'])
    bx = Boxes3D(n, dev)
    x = np.zeros((n, 8))
    x[:, 0], x[:, 1] = np.sin(g['Ry']), np.cos(g['Ry'])
    x[:, 2], x[:, 3], x[:, 4] = g['dimension'][:, 2], g['dimension'][:, 0], g['dimension'][:, 1]
    x[:, 5:8] = g['location']
    bx.x.copy_(torch.from_numpy(x).to(dev)); bx.status.fill_(0); bx.status[1] = -1
    proj, rect = kr.project_boxes_device(bx, np.tile(g['K'].reshape(1, 9), (n, 1)), topk=0)
    proj, rect = proj.cpu().numpy(), rect.cpu().numpy()
    for i in range(n):
        if i == 1:
            assert not proj[i].any() and not rect[i].any()
            continue
        snap = min(abs(np.sin(g['Ry'][i])), abs(np.cos(g['Ry'][i])))
        if abs(snap - 1e-3) < 1e-9:            # exactly on the snapping threshold: atan2(sin, cos) may land on either side
            continue
        np.testing.assert_allclose(proj[i], g['proj'][i], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(rect[i], np.concatenate([g['proj'][i][:8].min(0), g['proj'][i][:8].max(0)]), rtol=1e-9, atol=1e-7)
    with pytest.raises(ValueError):
        kr.project_boxes_device(bx, g['K'].reshape(1, 9), topk=7)


def test_config4_smoke_variant_bs32_full_size(dev):
    """BASELINE configs[4] at its full size (smoke head table, DLA-34, bs=32, 384x1280; PARITY UNPINNED: the branch's source
    is not in the reference snapshot).  Size-independent checks: two runs bit-identical, the device decode of the
    device's own logits equals the oracle's restatement of the decode on those logits (peaks bit-exact, closed-form boxes
    to fp64 libm accuracy) for a sample of images, slot bookkeeping consistent."""
    from oracle import smoke_ref
    from rtm3d_amd.model_utils import decode_smoke_slots
    bb = 'DLA-34'
    cfg = rtm3d_amd.kitti_config(bb)
    cfg.MODEL.HEAD_VARIANT = 'smoke'
    sd = weights.synth_state_dict(bb, 1, 'trained', heat_bias=-6.0, head_variant='smoke')
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    m.load_state_dict(sd)
    B = 32
    x = weights.synth_images(B, 384, 1280, seed=1234).to(dev)
    Knp = np.tile(weights.synth_intrinsics(), (B, 1))
    K = torch.as_tensor(Knp, device=dev)
    det, boxes, logits = m.detect3d(x, K)
    torch.cuda.synchronize()
    assert len(logits) == 2 and logits[0].shape == (B, 3, 96, 320) and logits[1].shape == (B, 8, 96, 320)
    n = det.n.cpu().numpy()
    assert n.sum() > 0 and n.max() <= 100 and int(n.sum()) == int((boxes.status >= 0).sum())
    keep = [l.clone() for l in logits]
    x8, cls0 = boxes.x.clone(), det.cls.clone()
    det2, boxes2, logits2 = m.detect3d(x, K)
    for a, b in zip(keep, logits2):
        assert torch.equal(a, b)
    assert torch.equal(x8, boxes2.x) and torch.equal(cls0, det2.cls)
    sample = [0, 17, 31]
    ref = smoke_ref.decode(keep[0][sample].cpu(), keep[1][sample].cpu(), Knp[sample], cfg.DETECTOR.dim_ref, 0.4, 100, 4.0)
    for j, b in enumerate(sample):
        if ref[j] is None:
            assert n[b] == 0
            continue
        k = len(ref[j]['cls'])
        assert n[b] == k
        np.testing.assert_array_equal(cls0[b * 100:b * 100 + k].cpu().numpy(), ref[j]['cls'])
        np.testing.assert_array_equal(det.score[b * 100:b * 100 + k].cpu().numpy(), ref[j]['score'])
        np.testing.assert_allclose(x8[b * 100:b * 100 + k].cpu().numpy(), ref[j]['x8'], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------------------ fp32 verification mode (SURVEY H2 ii)
# Model.forward_logits_fp32: the same plan on fp32 tensors (rtm3d_amd/verify.py).  Bars = 2 x the largest value measured on
# the MI355X (gpurun_out/measured_errors.json -> profiles/r02_logit_error.json, groups fp32_verify_*).
FP32_LOGIT_RTOL = 2e-5          # |logit - reference fp32 CPU logit| / max(1, max |logit|)
FP32_VERT_TOL_PX = 2e-3


@pytest.mark.parametrize('fname', E2E + E2E_NC)
def test_fp32_verify_logits_vs_reference_golden(dev, fname):
    """The verification executor reproduces the reference's fp32 CPU logits to fp32 round-off on every e2e fixture
    (three backbones, two sizes) - i.e. the recorded plan (graph wiring, BN folding, phases, composed 1x1s) is the
    reference's function, and what the fp16 tests above measure is storage rounding only."""
    g = load_golden(fname)
    bb = str(g['backbone'])
    B, H, W = [int(v) for v in g['shape']]
    nconv = int(g['header_num_conv']) if 'header_num_conv' in g else 2
    sd = weights.synth_state_dict(bb, int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']),
                                  header_num_conv=nconv)
    x = weights.synth_images(B, H, W, seed=int(g['img_seed']))
    m = make_model(bb, sd, nconv=nconv)
    logits = m.forward_logits_fp32(x.to(dev))
    errs = {'main_kf': _rel_err(logits[0].cpu().numpy(), g['logits_main_kf'])}
    for i, name in enumerate(['offset_fr_main', 'main_offset', 'vertex_offset'], 1):
        if 'logits_' + name in g:
            ref = g['logits_' + name]; got = logits[i].cpu().numpy()
        else:
            ref = g['logits_%s_s4' % name]; got = logits[i][:, :, ::4, ::4].cpu().numpy()
        errs[name] = _rel_err(got, ref)
    record_measurement('fp32_verify_logits_vs_reference_golden', fname, errs)
    for name, e in errs.items():
        assert e <= FP32_LOGIT_RTOL, (name, e)
    # every reference detection is found at the same rank with the same class and cell (no threshold margin needed
    # beyond fp32 round-off), vertices to FP32_VERT_TOL_PX
    d = m.inference(logits)
    n = g['det_n']
    vmax, smax, checked = 0.0, 0.0, 0
    thr_logit = float(np.log(0.4 / 0.6))
    for b in range(B):
        if n[b] == 0:
            continue
        rc, rs, rm, rv, _ = dets_from_golden(g, 'det_', b)
        margin = np.abs(np.log(rs.astype(np.float64) / (1.0 - rs.astype(np.float64))) - thr_logit)
        if (margin < 1e-3).any():
            continue                      # a score within fp32 noise of the threshold: membership is not defined
        assert d[0][b] is not None and len(d[0][b]) == n[b], (b, n[b])
        np.testing.assert_array_equal(to_np(d[0][b]), rc)
        np.testing.assert_array_equal(np.floor(to_np(d[2][b]) / 4), np.floor(rm / 4))
        smax = max(smax, float(np.abs(to_np(d[1][b]) - rs).max()))
        vmax = max(vmax, float(np.abs(to_np(d[3][b]) - rv).max()))
        checked += int(n[b])
    record_measurement('fp32_verify_detections_vs_reference_golden', fname, {'matched': checked, 'score_linf': smax, 'vertex_linf_px': vmax})
    assert checked >= 12 * B and vmax < FP32_VERT_TOL_PX and smax < 1.5e-5, (checked, vmax, smax)      # scores: 2 x the measured 7.1e-6


def test_fp32_verify_end_to_end_boxes_vs_reference(dev):
    """End to end in the verification mode, on objects the reference KEEPS: network (fp32 executor) -> decode2d ->
    decode3d_slots on the device against the reference's own Model.inference + optim_decode_bbox3d.  The cuboids of
    tests/golden/planted_small.npz are carried through the network additively: device logits + (planted - natural) of the
    reference-run logits, so each planted value on the device = exact projection + the device network's real error there.
    Bars: same detections, same kept set, vertices to fp32 round-off, box parameters: see the assertions."""
    name = 'planted_small'
    g = load_golden(name + '.npz')
    bg = load_golden(PLANTED_CASES[name][0])
    th, tk, K, arrs, _ = planted_inputs(name, bg)
    bb = str(bg['backbone'])
    B, H, W = [int(v) for v in bg['shape']]
    sd = weights.synth_state_dict(bb, int(bg['seed']), str(bg['style']), heat_bias=float(bg['heat_bias']), heat_gain=float(bg['heat_gain']))
    x = weights.synth_images(B, H, W, seed=int(bg['img_seed']))
    m = make_model(bb, sd, th, tk)
    natural = [bg['logits_' + n] for n in ('main_kf', 'offset_fr_main', 'main_offset', 'vertex_offset')]
    out = {}
    for mode in ('fp32', 'fp16'):
        lg = m.forward_logits_fp32(x.to(dev)) if mode == 'fp32' else m.forward_logits(x.to(dev))
        lg = [l + torch.from_numpy(p - nat).to(dev) for l, p, nat in zip(lg, arrs, natural)]
        det = m.decode2d(lg)
        dim_ref = rtm3d_amd.kitti_config().DETECTOR.dim_ref
        boxes = rtm3d_amd.model_utils.decode3d_slots(det, torch.as_tensor(np.tile(K, (B, 1)), device=dev), dim_ref, [0, -0.5, 20])
        torch.cuda.synchronize()
        xs, fs = boxes.x.cpu().numpy(), boxes.fun.cpu().numpy()
        st = {'kept_ref': 0, 'kept_both': 0, 'box_linf': 0.0, 'vert_linf_px': 0.0, 'same_detections': True}
        per_obj = []
        for b in range(B):
            nb = int(g['det_n'][b])
            rc, rs, rm, rv, _ = dets_from_golden(g, 'det_', b)
            same = int(det.n[b].item()) == nb and np.array_equal(det.cls[b * tk:b * tk + nb].cpu().numpy(), rc) \
                and np.array_equal(np.floor(det.mproj[b * tk:b * tk + nb].cpu().numpy() / 4), np.floor(rm / 4))
            st['same_detections'] = st['same_detections'] and bool(same)
            if not same:
                continue
            st['vert_linf_px'] = max(st['vert_linf_px'], float(np.abs(det.verts[b * tk:b * tk + nb].cpu().numpy() - rv).max()))
            rx, rf = g['d3_raw_x_%d' % b], g['d3_raw_fun_%d' % b]
            kept = rf < 0.1
            kd = fs[b * tk:b * tk + nb] < 0.1
            st['kept_ref'] += int(kept.sum())
            both = kept & kd
            st['kept_both'] += int(both.sum())
            if both.any():
                dv = np.abs(xs[b * tk:b * tk + nb][both] - rx[both])
                per_obj.append(dv)
                st['box_linf'] = max(st['box_linf'], float(dv.max()))
        if per_obj:
            dv = np.concatenate(per_obj)
            st['box_median'] = float(np.median(dv.max(1)))
            st['boxes_within_1e-4'] = int((dv.max(1) <= 1e-4).sum())
            st['box_linf_per_param'] = [float(v) for v in dv.max(0)]        # [sin, cos, l, h, w, X, Y, Z]
        out[mode] = st
    record_measurement('fp32_verify_end_to_end_boxes', name, out)
    f = out['fp32']
    assert f['same_detections'] and f['kept_ref'] >= 20 and f['kept_both'] == f['kept_ref'], f
    assert f['vert_linf_px'] < FP32_VERT_TOL_PX, f
    # north_star's 1e-4 holds for the typical box; the maximum belongs to the objects whose L-BFGS-B run stops one
    # iteration apart: the REFERENCE itself moves l / w of such an object by up to 2.1e-3 when its input vertices change by
    # 3e-5 px (tools/solver_sensitivity.py -> profiles/r02_solver_sensitivity.txt), which is this mode's vertex error.
    # Measured: median 6e-6, maximum 8.2e-4; the reference under 3e-5 px of input noise: 2.1e-3 (bar = 2 x that).
    assert f['box_median'] <= 1e-4 and f['box_linf'] <= 4.2e-3, f
    assert f['boxes_within_1e-4'] >= int(0.75 * f['kept_ref']), f


def test_fp16_range_report_names_the_overflowing_tensor(dev):
    """the product path stores activations as fp16 with no clamp.  Model.check_range (fp32 verification
    executor) reports, per written tensor, max |x| against 65504.  A healthy state dict has head-room everywhere; with ONE
    backbone BatchNorm scaled so that level3's first block leaves the fp16 range the report names that tensor (and the ones
    downstream), strict=True raises naming the first one in plan order, and the PRODUCT path's logits for that input are silently
    wrong (see below); non-finite logits reach decode2d as the NaN / Inf cells the decode tests cover."""
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 3, 'trained', heat_bias=-3.0)
    x = weights.synth_images(1, 128, 256, seed=5).to(dev)
    m = make_model(bb, sd)
    rows = m.check_range(x, strict=True)
    acts = [r for r in rows if r['what'] == 'activation']
    assert len(acts) >= 40 and not any(r['overflow'] for r in rows)
    assert min(r['headroom'] for r in rows) > 50, rows[0]                 # the synthetic checkpoints sit far inside the range
    #  the report also covers what the level rewrites store: composed neck taps / summed biases of the recorded ops,
    # and it says which activation rows the realized plan no longer materialises (the neck's `up` maps, s2d-only features)
    realized = [r for r in rows if r['what'] == 'weight (realized)']
    rplan = m._plan_for(1, 128, 256, dev)
    assert len(realized) == len(rplan.weight_ranges) > 20 and any(r['materialised'] for r in acts)
    ghost = [r for r in acts if not r['materialised']]
    assert bool(ghost) == bool(rplan.unwritten)                          # rows the realized plan no longer writes are tagged as such
    if any('+kfpn_proj' in n for n in rplan.op_names):                   # (the neck fold applies from a certain map size on)
        assert any('kfpn_up' in r['op'] and 'kfpn_proj' in r['op'] for r in realized) and any('kfpn_up' in r['op'] for r in ghost)
    record_measurement('fp16_range', 'healthy', {'largest': rows[0]['max_abs'], 'tensor': rows[0]['tensor'], 'op': rows[0]['op']})
    m.release_verify()
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2['backbone.level3.tree1.tree1.norm1.weight'] *= 3.0e5
    m2 = make_model(bb, sd2)
    rows2 = m2.check_range(x)
    bad = [r for r in rows2 if r['overflow']]
    assert bad and any(r['op'] == 'backbone.level3.tree1.tree1.conv1' for r in bad), [(r['op'], r['max_abs']) for r in rows2[:5]]
    with pytest.raises(OverflowError, match='backbone.level3'):
        m2.check_range(x, strict=True)
    # what the PRODUCT path does with that checkpoint: nothing announces the overflow.  The stores round the out-of-range values
    # to +-inf; an inf times weights of both signs sums to NaN; and the kernels' ReLU is v_pk_max_f16, an IEEE maxNum, which
    # returns 0 for a NaN operand (torch.relu in the reference would propagate it) - so the logits may come out as non-finite
    # values OR as finite garbage.  Either way they are far from the fp32 run's, which is why check_range exists.
    ref32 = [l.clone() for l in m2.forward_logits_fp32(x)]
    logits = m2.forward_logits(x)
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(l).all()) for l in logits)
    gross = max(float((a - b).abs().max() / b.abs().max().clamp_min(1.0)) for a, b in zip(logits, ref32)) if finite else float('inf')
    record_measurement('fp16_range', 'overflowing', {'first_bad_op': sorted(bad, key=lambda r: r['order'])[0]['op'], 'tensors_over': len(bad),
                                                     'product_logits_finite': finite, 'product_vs_fp32_rel_err': gross if finite else -1.0})
    assert (not finite) or gross > 0.5, gross
    d = m2.inference([l.clone() for l in logits])                        # ... which the decode handles (no crash, no NaN-scored detection)
    if d[0][0] is not None:
        assert bool(torch.isfinite(d[1][0]).all())


def test_detect_py_shaped_loop_on_the_kitti416_fixture(dev):
    """the reference's detect.py loop (detect.py:56-74) with this package, on the real-KITTI letterbox
    shape 1 x 3 x 416 x 1280 (reference-run fixture e2e_dla34_kitti416.npz): `Model.detect` gives the five lists, every reference
    detection safely above the threshold is found with its vertices within VERT_TOL_PX, and `optim_decode_bbox3d` on those lists
    returns a ParamList with the reference's fields whose kept set equals the set the fixture's raw solver states keep when the
    detections match one to one."""
    from rtm3d_amd import model_utils
    g = load_golden('e2e_dla34_kitti416.npz')
    bb = str(g['backbone'])
    B, H, W = [int(v) for v in g['shape']]
    assert (B, H, W) == (1, 416, 1280)
    sd = weights.synth_state_dict(bb, int(g['seed']), str(g['style']), heat_bias=float(g['heat_bias']), heat_gain=float(g['heat_gain']))
    cfg = rtm3d_amd.kitti_config(bb)
    model = rtm3d_amd.create_model(cfg).to(dev).eval()
    model.load_state_dict(sd)
    imgs = weights.synth_images(B, H, W, seed=int(g['img_seed']))
    K = np.asarray(g['K'], np.float64)
    preds = model.detect(imgs.to(dev))
    clses, m_scores, m_projs, v_projs_regress, bboxes_2d = preds
    assert clses[0] is not None
    out = model_utils.optim_decode_bbox3d(clses[0].cpu().numpy(), v_projs_regress[0].cpu().numpy(), K, cfg.DETECTOR.dim_ref, [0, -0.5, 20])
    for f in ('class', 'Ry', 'dimension', 'location', 'K'):
        assert out.has_field(f)
    assert np.asarray(out.get_field('dimension')).reshape(-1, 3).shape[1] == 3 and np.asarray(out.get_field('K')).reshape(-1, 9).shape[1] == 9
    # detections against the reference's (margin rule of the golden tests)
    rc, rs, rm, rv, _ = dets_from_golden(g, 'det_', 0)
    tol = HM_RTOL * max(1.0, np.abs(g['logits_main_kf']).max())
    sure = np.abs(np.log(rs.astype(np.float64) / (1.0 - rs.astype(np.float64))) - float(np.log(0.4 / 0.6))) > tol
    got = {(int(c), int(mx // 4), int(my // 4)): v for c, (mx, my), v in
           zip(clses[0].cpu().numpy(), m_projs[0].cpu().numpy(), v_projs_regress[0].cpu().numpy())}
    checked = 0
    for c, mp, v, ok in zip(rc, rm, rv, sure):
        if ok:
            key = (int(c), int(mp[0] // 4), int(mp[1] // 4))
            assert key in got, key
            assert np.abs(got[key] - v).max() < VERT_TOL_PX
            checked += 1
    assert checked >= 12
    # the kept set: the reference keeps none of this synthetic-weight frame's detections (fun >= 0.1 for all), and so does the device
    if len(rc) == len(clses[0]):
        assert len(out.get_field('class')) == int((g['d3_raw_fun_0'] < 0.1).sum())
