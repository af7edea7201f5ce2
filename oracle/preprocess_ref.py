"""CPU restatement (numpy) of the reference's letterbox + normalisation step.

TEST INFRASTRUCTURE ONLY.  Follows datasets/dataset_reader.py:175-195 (`_apply_padding`: canvas filled with
`cv2.mean(img)[:3]` cast to uint8, image centred, K shifted) and preprocess/transforms.py:110-120,
312-322 (`Normalize` in float64 with float32 mean/std, `ToTensor` -> float32, `ToNCHW`).
Pinned by tests/golden/preprocess_cases.npz: Normalize/ToTensor/ToNCHW and `_apply_padding` were run from
the reference's own code (with `cv2.mean` provided by a numpy stub, since OpenCV is absent here).
"""
import numpy as np


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
