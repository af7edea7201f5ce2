#!/usr/bin/env python3
"""Golden vectors for the input-pipeline step, produced by RUNNING the reference's own
code: `DatasetReader._apply_padding` (datasets/dataset_reader.py:175-195) and
`Normalize`/`ToTensor`/`ToNCHW` (preprocess/transforms.py).  Run only in the build container:

    cd /tmp && python -B /root/repo/tests/golden/make_golden_preprocess.py

Modules the image lacks are stubbed in sys.modules (no reference source is copied): torchvision,
albumentations, the external KITTI devkit, and cv2 - whose single function used on this path, `cv2.mean`,
is provided as the per-channel arithmetic mean in float64 (OpenCV's documented definition).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


tv = _stub('torchvision'); tvt = _stub('torchvision.transforms', ToTensor=lambda: None); tv.transforms = tvt
_stub('albumentations')
cv2 = _stub('cv2', mean=lambda img: tuple(np.asarray(img, np.float64).reshape(-1, img.shape[2]).mean(axis=0)) + (0.0,) * (4 - img.shape[2]),
            INTER_LINEAR=1,
            # size bookkeeping only: lets the reference's own Resize.__call__ run so that its dsize rule and the K scaling of
            # ToPercentCoords/ToAbsoluteCoords are pinned; the interpolated pixel values cannot be (OpenCV is absent)
            resize=lambda src, dsize, interpolation: np.zeros((dsize[1], dsize[0], src.shape[2]), src.dtype))
for n in ('datasets.data', 'datasets.data.kitti', 'datasets.data.kitti.devkit_object', 'datasets.data.kitti.devkit_object.utils'):
    _stub(n)
sys.modules['datasets.data.kitti.devkit_object'].utils = sys.modules['datasets.data.kitti.devkit_object.utils']

from preprocess import transforms as ref_t          # noqa: E402 (reference)
from datasets.dataset_reader import DatasetReader   # noqa: E402 (reference)
from utils.ParamList import ParamList               # noqa: E402 (reference)

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]   # models/configs/detault.py:25-26
rng = np.random.Generator(np.random.PCG64(42))
out = {'mean': np.array(MEAN), 'std': np.array(STD)}
cases = [(370, 1224, 384, 1280), (96, 320, 128, 352), (33, 57, 64, 64), (64, 64, 64, 64)]
out['cases'] = np.array(cases)
for i, (h, w, H, W) in enumerate(cases):
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    if i == 2:
        img[..., 1] = 255                         # saturated channel: mean exactly 255
    tgt = ParamList((w, h))
    tgt.add_field('bbox', np.zeros((1, 4)))
    tgt.add_field('K', np.array([[700., 0, 600, 0, 700, 180, 0, 0, 1]]))
    fake = types.SimpleNamespace(_img_size=(W, H))
    nimg, tgt = DatasetReader._apply_padding(fake, [img], [tgt])
    norm = {'mean_rgb': np.array(MEAN, np.float32).reshape((1, 1, 3)), 'std_rgb': np.array(STD, np.float32).reshape((1, 1, 3))}
    x, _ = ref_t.Normalize()(np.ascontiguousarray(nimg), None, **norm)
    x, _ = ref_t.ToTensor()(x, None)
    x, _ = ref_t.ToNCHW()(x, None)
    out['img_seed'] = 42
    out['probe_%d' % i] = img[:2, :4].copy()
    out['canvas_corner_%d' % i] = nimg[0, 0].copy()
    out['K_%d' % i] = tgt.get_field('K')
    x = x.numpy()
    # store the full tensor for the small cases, a checksum + samples for the large one
    if H * W <= 128 * 352:
        out['x_%d' % i] = x
    else:
        out['x_%d_sub' % i] = x[:, ::7, ::11].copy()
        out['x_%d_sum' % i] = x.astype(np.float64).sum(axis=(1, 2))
# Resize bookkeeping through the reference's TestTransform chain (preprocess/data_preprocess.py:35-44):
# ToPercentCoords -> Resize(size) -> ToAbsoluteCoords on (image shape, K)
rs_cases = [(375, 1242, 1280), (370, 1224, 1280), (376, 1241, 1280), (100, 57, 64), (480, 640, 1280), (1, 1, 32)]
out['rs_cases'] = np.array(rs_cases)
for i, (h, w, size) in enumerate(rs_cases):
    img = np.zeros((h, w, 3), np.uint8)
    tgt = ParamList((w, h))
    tgt.add_field('K', np.array([[721.5377, 0, 609.5593, 0, 721.5377, 172.854, 0, 0, 1]]))
    for tr in (ref_t.ToPercentCoords(), ref_t.Resize(size), ref_t.ToAbsoluteCoords()):
        img, tgt = tr(img, tgt)
    out['rs_size_%d' % i] = np.array(img.shape[:2])
    out['rs_K_%d' % i] = tgt.get_field('K')
np.savez_compressed(os.path.join(HERE, 'preprocess_cases.npz'), **out)
print('ok', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})
