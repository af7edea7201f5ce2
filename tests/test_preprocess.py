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
