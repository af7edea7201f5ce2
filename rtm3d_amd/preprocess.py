"""Input pipeline step in front of the hot path on the GPU.

``letterbox_normalize(images, size, mean, std)`` takes already resized uint8 HWC images (what
``TestTransform``'s ``Resize`` hands to ``DatasetReader._apply_padding``), pads each one centred into the
network canvas with its own mean colour (datasets/dataset_reader.py:175-195) and applies
``Normalize -> ToTensor -> ToNCHW`` (preprocess/transforms.py:110-120, 312-322).  ``adjust_K`` shifts the
principal point like dataset_reader.py:189-193.  The bilinear ``cv2.resize`` itself is not part of this
step (OpenCV's fixed-point resize cannot be pinned here: cv2 is absent from the build image).
"""
import ctypes

import numpy as np
import torch

from . import _lib


def normalize_lut(mean, std):
    """float32((v / 255. - mean[c]) / std[c]) for v in 0..255, evaluated in float64 with float32 mean/std
    exactly like ``Normalize`` + ``ToTensor`` (transforms.py:110-120, 312-317)."""
    v = np.arange(256, dtype=np.float64)[None, :] / 255.
    m = np.asarray(mean, np.float32).reshape(3, 1)
    s = np.asarray(std, np.float32).reshape(3, 1)
    return np.ascontiguousarray(((v - m) / s).astype(np.float32))


def letterbox_normalize(images, size, mean, std, out=None):
    """images: list of uint8 CUDA tensors (h, w, 3); size = (H, W).  Returns ((B,3,H,W) fp32 CUDA tensor,
    [(pad_w, pad_h)] per image)."""
    lib = _lib.load()
    H, W = int(size[0]), int(size[1])
    if not images:
        raise ValueError('no images')
    dev = images[0].device
    if dev.type != 'cuda':
        raise RuntimeError('rtm3d_amd.preprocess needs CUDA (ROCm) tensors; there is no CPU path')
    B = len(images)
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        lut = torch.as_tensor(normalize_lut(mean, std), device=dev)
        sums = torch.zeros(B, 3, dtype=torch.int64, device=dev)
        pads = []
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for b, img in enumerate(images):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
                raise ValueError('expected uint8 (h, w, 3) images')
            img = img.contiguous()
            h, w = int(img.shape[0]), int(img.shape[1])
            _lib.check(lib.rtm3d_preprocess(stream, img.data_ptr(), h, w, out[b].data_ptr(), H, W, lut.data_ptr(), sums[b].data_ptr()),
                       'preprocess')
            pads.append(((W - w) // 2, (H - h) // 2))
    return out, pads


def adjust_K(K, pad_w, pad_h):
    """K: (..., 9) row-major intrinsics; cx += pad_w, cy += pad_h (datasets/dataset_reader.py:189-193)."""
    K = np.array(K, dtype=np.float64, copy=True).reshape(-1, 9)
    K[:, 2] += pad_w
    K[:, 5] += pad_h
    return K
