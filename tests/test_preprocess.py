"""Input-pipeline step: oracle vs the reference-generated golden vectors (CPU) and the
HIP kernel vs both (GPU), bit-exact (a 3x256 look-up of float64-evaluated, float32-rounded values)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref
from tests.util import load_golden


def _inputs(g):
    rng = np.random.Generator(np.random.PCG64(int(g['img_seed'])))
    imgs = []
    for i, (h, w, H, W) in enumerate(g['cases']):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        if i == 2:
            img[..., 1] = 255
        np.testing.assert_array_equal(img[:2, :4], g['probe_%d' % i])
        imgs.append(img)
    return imgs


def _check(x, g, i):
    if 'x_%d' % i in g:
        np.testing.assert_array_equal(x, g['x_%d' % i])
    else:
        np.testing.assert_array_equal(x[:, ::7, ::11], g['x_%d_sub' % i])
        np.testing.assert_allclose(x.astype(np.float64).sum(axis=(1, 2)), g['x_%d_sum' % i], rtol=0, atol=1e-6)


def test_oracle_preprocess_matches_reference():
    g = load_golden('preprocess_cases.npz')
    for i, (img, (h, w, H, W)) in enumerate(zip(_inputs(g), g['cases'])):
        x, pw, ph = preprocess_ref.letterbox_normalize(img, (H, W), g['mean'], g['std'])
        _check(x, g, i)
        np.testing.assert_array_equal(g['K_%d' % i], np.array([[700., 0, 600 + pw, 0, 700, 180 + ph, 0, 0, 1]]))


@pytest.mark.gpu
def test_hip_preprocess_bit_exact():
    from rtm3d_amd import preprocess
    g = load_golden('preprocess_cases.npz')
    imgs = _inputs(g)
    for i, (img, (h, w, H, W)) in enumerate(zip(imgs, g['cases'])):
        out, pads = preprocess.letterbox_normalize([torch.from_numpy(img).cuda()], (int(H), int(W)), g['mean'], g['std'])
        _check(out[0].cpu().numpy(), g, i)
        K = preprocess.adjust_K(np.array([700., 0, 600, 0, 700, 180, 0, 0, 1]), *pads[0])
        np.testing.assert_array_equal(K, g['K_%d' % i])
    # batch of ragged images in one call + fresh random image vs the oracle
    rng = np.random.Generator(np.random.PCG64(5))
    batch = [rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8) for hh, ww in ((96, 300), (100, 320), (1, 1))]
    out, pads = preprocess.letterbox_normalize([torch.from_numpy(b).cuda() for b in batch], (128, 320), g['mean'], g['std'])
    for b, img in enumerate(batch):
        ref, pw, ph = preprocess_ref.letterbox_normalize(img, (128, 320), g['mean'], g['std'])
        np.testing.assert_array_equal(out[b].cpu().numpy(), ref)
        assert pads[b] == (pw, ph)
    with pytest.raises(RuntimeError):
        preprocess.letterbox_normalize([torch.zeros(200, 10, 3, dtype=torch.uint8).cuda()], (128, 320), g['mean'], g['std'])


def test_resize_bookkeeping_matches_reference():
    """Resize's dsize rule and the K scaling of ToPercentCoords/ToAbsoluteCoords, against vectors from the reference's own
    TestTransform chain (only the interpolated pixel values are unpinned: OpenCV is absent)."""
    from rtm3d_amd import preprocess
    g = load_golden('preprocess_cases.npz')
    K0 = np.array([721.5377, 0, 609.5593, 0, 721.5377, 172.854, 0, 0, 1])
    for i, (h, w, size) in enumerate(g['rs_cases']):
        for mod in (preprocess_ref, preprocess):
            nh, nw = mod.resized_size(int(h), int(w), int(size))
            np.testing.assert_array_equal([nh, nw], g['rs_size_%d' % i])
        np.testing.assert_array_equal(preprocess_ref.test_transform_K(K0, (h, w), (nh, nw)), g['rs_K_%d' % i])
        np.testing.assert_array_equal(preprocess.resize_K(K0, (h, w), (nh, nw)), g['rs_K_%d' % i])


def test_oracle_resize_properties():
    """OpenCV-style fixed-point bilinear restatement (PARITY UNPINNED): identity at equal size, exact on constant
    images, exact 2x down-sampling of a 2-periodic pattern, monotone on a ramp, and within 1 LSB of float bilinear."""
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    np.testing.assert_array_equal(preprocess_ref.resize_bilinear_u8(img, (37, 53)), img)
    const = np.full((20, 30, 3), 137, np.uint8)
    assert (preprocess_ref.resize_bilinear_u8(const, (33, 47)) == 137).all()
    chk = np.zeros((8, 8, 3), np.uint8); chk[:, 1::2] = 200
    assert (preprocess_ref.resize_bilinear_u8(chk, (4, 4)) == 100).all()          # each output = mean of a (0, 200) pair
    ramp = np.repeat(np.arange(0, 250, 5, dtype=np.uint8)[None, :, None], 9, 0).repeat(3, 2)
    up = preprocess_ref.resize_bilinear_u8(ramp, (9, 123)).astype(int)
    assert (np.diff(up[0, :, 0]) >= 0).all()
    big = preprocess_ref.resize_bilinear_u8(img, (61, 99)).astype(np.float64)
    ys = np.clip((np.arange(61) + 0.5) * 37 / 61 - 0.5, 0, 36); xs = np.clip((np.arange(99) + 0.5) * 53 / 99 - 0.5, 0, 52)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int); y1 = np.minimum(y0 + 1, 36); x1 = np.minimum(x0 + 1, 52)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(big - ref).max() <= 1.0


@pytest.mark.gpu
def test_hip_preprocess_batch_vs_oracle_and_network_input():
    """rtm3d_preprocess_batch: (1) no-resize path bit-identical to the reference-pinned single-image kernel; (2) with the
    Resize in front, bit-identical to the oracle's restatement (parity unpinned); (3) the fp16 NHWC4 output written straight
    into the plan's input tensor gives bit-identical logits to feeding the fp32 batch through Model.forward."""
    import rtm3d_amd
    from rtm3d_amd import preprocess, weights
    g = load_golden('preprocess_cases.npz')
    mean, std = g['mean'], g['std']
    rng = np.random.Generator(np.random.PCG64(9))
    sizes = [(96, 300), (100, 320), (1, 1), (128, 320), (77, 123)]
    batch = [rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8) for hh, ww in sizes]
    dev_imgs = [torch.from_numpy(b).cuda() for b in batch]
    # (1) already resized images: equals letterbox_normalize (which is pinned by reference-run vectors)
    single, pads1 = preprocess.letterbox_normalize(dev_imgs, (128, 320), mean, std)
    out, pads, rhw = preprocess.preprocess_batch(dev_imgs, (128, 320), mean, std)
    torch.cuda.synchronize()
    assert torch.equal(out, single) and pads == pads1 and rhw == sizes
    # (2) Resize to longest side 320 (the reference's rule), then letterbox
    originals = [rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8) for hh, ww in ((375, 1242), (200, 640), (640, 200), (31, 17))]
    out, pads, rhw = preprocess.preprocess_batch([torch.from_numpy(b).cuda() for b in originals], (320, 320), mean, std, resize_to=320)
    for b, img in enumerate(originals):
        nh, nw = preprocess_ref.resized_size(img.shape[0], img.shape[1], 320)
        assert rhw[b] == (nh, nw)
        small = preprocess_ref.resize_bilinear_u8(img, (nh, nw))
        ref, pw, ph = preprocess_ref.letterbox_normalize(small, (320, 320), mean, std)
        np.testing.assert_array_equal(out[b].cpu().numpy(), ref)
        assert pads[b] == (pw, ph)
    # (3) straight into the network's input tensor
    bb = 'RESNET-18'
    cfg = rtm3d_amd.kitti_config(bb)
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    imgs = [rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8) for hh, ww in ((90, 250), (128, 256), (64, 100))]
    d = [torch.from_numpy(b).cuda() for b in imgs]
    x32, _, _ = preprocess.preprocess_batch(d, (128, 256), mean, std)
    ref_logits = m.forward_logits(x32)
    _, pads, _ = preprocess.preprocess_batch(d, (128, 256), mean, std, model=m)
    got = m.forward_logits(None, preloaded=(3, 128, 256))
    torch.cuda.synchronize()
    for a, b in zip(got, ref_logits):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        preprocess.preprocess_batch([torch.zeros(200, 10, 3, dtype=torch.uint8).cuda()], (128, 320), mean, std)


@pytest.mark.gpu
def test_pipeline_submit_uint8_equals_fp32_feed():
    """Detect3DPipeline.submit_uint8 (camera images -> Resize + letterbox + normalise -> plan, n1 in front of the path) gives
    the records of the same batch fed as the reference's fp32 NCHW tensor, bit for bit, over several pipelined steps."""
    import rtm3d_amd
    from rtm3d_amd import preprocess, weights
    from rtm3d_amd.pipeline import Detect3DPipeline
    bb = 'RESNET-18'
    cfg = rtm3d_amd.kitti_config(bb)
    dev = torch.device('cuda', 0)
    m = rtm3d_amd.create_model(cfg).to(dev).eval()
    m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    rng = np.random.Generator(np.random.PCG64(10))
    H, W, B = 128, 256, 3
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), dtype=torch.float64, device=dev)
    pipe_a = Detect3DPipeline(m, B, dev, gather=False)
    pipe_b = Detect3DPipeline(m, B, dev, gather=False)
    for step in range(3):
        imgs = [torch.from_numpy(rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8)).to(dev) for hh, ww in ((180, 500), (100, 256), (120, 300))]
        x32, _, _ = preprocess.preprocess_batch(imgs, (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, resize_to=256)
        ra = pipe_a.results(pipe_a.submit(x32, K)).clone()
        rb = pipe_b.results(pipe_b.submit_uint8(imgs, K, (H, W), resize_to=256)).clone()
        torch.cuda.synchronize()
        assert torch.equal(ra, rb) and float(ra[:, :, 31].sum()) > 0
    with pytest.raises(ValueError):
        pipe_b.submit_uint8(imgs[:2], K, (H, W), resize_to=256)
    # the same with the peaks-only regression heads: the uint8 feed must fill the input tensor of the plan that is replayed
    # (the heat-map-only plan owns its own workspace)
    pipe_c = Detect3DPipeline(m, B, dev, gather=False, sparse_heads=True)
    pipe_d = Detect3DPipeline(m, B, dev, gather=False, sparse_heads=True)
    for step in range(2):
        imgs = [torch.from_numpy(rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8)).to(dev) for hh, ww in ((180, 500), (100, 256), (120, 300))]
        x32, _, _ = preprocess.preprocess_batch(imgs, (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, resize_to=256)
        rc = pipe_c.results(pipe_c.submit(x32, K)).clone()
        rd = pipe_d.results(pipe_d.submit_uint8(imgs, K, (H, W), resize_to=256)).clone()
        torch.cuda.synchronize()
        assert torch.equal(rc, rd) and float(rc[:, :, 31].sum()) > 0
