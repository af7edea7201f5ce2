"""CPU restatement (numpy) of the reference's letterbox + normalisation step.

TEST INFRASTRUCTURE ONLY.  Follows datasets/dataset_reader.py:175-195 (`_apply_padding`: canvas filled with
`cv2.mean(img)[:3]` cast to uint8, image centred, K shifted) and preprocess/transforms.py:110-120,
312-322 (`Normalize` in float64 with float32 mean/std, `ToTensor` -> float32, `ToNCHW`).
Pinned by tests/golden/preprocess_cases.npz: Normalize/ToTensor/ToNCHW and `_apply_padding` were run from
the reference's own code (with `cv2.mean` provided by a numpy stub, since OpenCV is absent here).

`resize_bilinear_u8` restates `cv2.resize(..., interpolation=cv2.INTER_LINEAR)` for 8-bit 3-channel images from
OpenCV's published algorithm (imgproc/resize.cpp: 11-bit fixed-point coefficients, int32 horizontal pass,
`(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2` vertical pass).  OpenCV is a third-party dependency of
the reference (unpinned; preprocess/transforms.py:480-495 is the call site) that is NOT installed in this image, and the
reference holds no fixture for it: PARITY UNPINNED for the resize step.
"""
import numpy as np


def _resize_coef(dsize, ssize):
    scale = np.float64(ssize) / np.float64(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= ssize - 1
    f[hi] = 0; s[hi] = ssize - 1
    s1 = np.minimum(s + 1, ssize - 1)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)       # cvRound: round half to even
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, s1, c0, c1


def resize_bilinear_u8(img, new_hw):
    """img (h, w, 3) uint8 -> (h', w', 3) uint8; equal sizes return a copy (as OpenCV does)."""
    h, w = img.shape[:2]
    nh, nw = int(new_hw[0]), int(new_hw[1])
    if (nh, nw) == (h, w):
        return img.copy()
    x0, x1, a0, a1 = _resize_coef(nw, w)
    y0, y1, b0, b1 = _resize_coef(nh, h)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # (h, nw, 3) int
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def resized_size(h, w, size):
    """preprocess/transforms.py:484-490 (int size): rate = size / max(h, w); (int(h * rate), int(w * rate))."""
    rate = size / max(h, w)
    return int(h * rate), int(w * rate)


def test_transform_K(K, hw, new_hw):
    """ToPercentCoords / ToAbsoluteCoords on K (preprocess/transforms.py:146-176)."""
    K = np.array(K, np.float64).reshape(-1, 9).copy()
    K[:, :3] /= hw[1]; K[:, 3:6] /= hw[0]
    K[:, :3] *= new_hw[1]; K[:, 3:6] *= new_hw[0]
    return K


def apply_padding(img, size_wh):
    sw, sh = size_wh
    h, w, c = img.shape
    mean_rgb = img.reshape(-1, c).astype(np.float64).mean(axis=0)[:3]        # cv2.mean: arithmetic mean per channel
    nimg = np.full((sh, sw, c), mean_rgb, dtype=np.uint8)
    pad_w = int(sw - w) // 2
    pad_h = int(sh - h) // 2
    nimg[pad_h:pad_h + h, pad_w:pad_w + w] = img
    return nimg, pad_w, pad_h


def normalize_to_nchw(img_u8, mean, std):
    mean = np.array(mean, np.float32).reshape((1, 1, 3))
    std = np.array(std, np.float32).reshape((1, 1, 3))
    img = img_u8 / 255.
    img -= mean
    img /= std
    return np.ascontiguousarray(img.astype(np.float32).transpose(2, 0, 1))


def letterbox_normalize(img_u8, size_hw, mean, std):
    nimg, pad_w, pad_h = apply_padding(img_u8, (size_hw[1], size_hw[0]))
    return normalize_to_nchw(nimg, mean, std), pad_w, pad_h
