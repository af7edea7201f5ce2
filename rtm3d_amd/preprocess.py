"""Input pipeline step in front of the hot path on the GPU.

``letterbox_normalize(images, size, mean, std)`` takes already resized uint8 HWC images (what
``TestTransform``'s ``Resize`` hands to ``DatasetReader._apply_padding``), pads each one centred into the
network canvas with its own mean colour (datasets/dataset_reader.py:175-195) and applies
``Normalize -> ToTensor -> ToNCHW`` (preprocess/transforms.py:110-120, 312-322).  ``adjust_K`` shifts the
principal point like dataset_reader.py:189-193.

``preprocess_batch`` is the whole step for a batch of ragged ORIGINAL images in two kernel launches: ``Resize``
(transforms.py:480-495: longest side -> ``resize_to``, ``cv2.resize(INTER_LINEAR)``) + letterbox + normalise, writing
either the reference's fp32 NCHW batch or directly the network's fp16 NHWC4 input tensor (``Model.forward_logits`` is
then called with ``preloaded=True``).  ``resize_K`` follows ToPercentCoords -> Resize -> ToAbsoluteCoords
(transforms.py:146-176).  The resize restates OpenCV's published fixed-point bilinear algorithm; OpenCV is absent from
the build image and the reference holds no fixture for it, so that one step is PARITY UNPINNED (everything else here is
bit-exact against vectors produced by the reference's own code).
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


_LUTS = {}


def device_luts(mean, std, dev):
    """(fp32 LUT, fp16 LUT) on the device, cached per (mean, std, device): building them per call would put a pageable
    host->device copy (which blocks the host and serialises streams) in front of every batch."""
    key = (tuple(float(v) for v in mean), tuple(float(v) for v in std), dev.index)
    hit = _LUTS.get(key)
    if hit is None:
        lut = normalize_lut(mean, std)
        with torch.cuda.device(dev):
            hit = (torch.as_tensor(lut, device=dev), torch.as_tensor(lut.astype(np.float16), device=dev))   # round-to-nearest-even, as the device cast
        _LUTS[key] = hit
    return hit


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


def resized_size(h, w, resize_to):
    """transforms.Resize with an int size (preprocess/transforms.py:484-490): rate = size / max(h, w),
    dsize = (int(w * rate), int(h * rate)).  Returns (h', w')."""
    rate = resize_to / max(h, w)
    return int(h * rate), int(w * rate)


def resize_K(K, hw, new_hw):
    """ToPercentCoords -> Resize -> ToAbsoluteCoords on the intrinsics (transforms.py:146-176):
    K[:, :3] /= w; K[:, 3:6] /= h; then K[:, :3] *= w'; K[:, 3:6] *= h'."""
    K = np.array(K, dtype=np.float64, copy=True).reshape(-1, 9)
    K[:, :3] /= hw[1]
    K[:, 3:6] /= hw[0]
    K[:, :3] *= new_hw[1]
    K[:, 3:6] *= new_hw[0]
    return K


def preprocess_batch(images, size, mean, std, resize_to=None, out=None, model=None, heads='dense'):
    """images: list of uint8 CUDA tensors (h, w, 3), any sizes.  size = (H, W) network canvas.
    resize_to: None (images are already resized) or the reference's INPUT_SIZE (longest side after Resize).
    model given: write straight into that model's fp16 NHWC4 input tensor for batch (B, H, W) - call
    ``model.forward_logits(None, preloaded=(B, H, W), heads=heads)`` next (heads: the plan that forward will replay);
    otherwise returns the fp32 (B,3,H,W) batch.
    Returns (out or None, [(pad_w, pad_h)], [(h', w')])."""
    lib = _lib.load()
    H, W = int(size[0]), int(size[1])
    if not images:
        raise ValueError('no images')
    dev = images[0].device
    if dev.type != 'cuda':
        raise RuntimeError('rtm3d_amd.preprocess needs CUDA (ROCm) tensors; there is no CPU path')
    B = len(images)
    imgs = []
    for img in images:
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError('expected uint8 (h, w, 3) images')
        imgs.append(img.contiguous())
    hw = np.array([[int(i.shape[0]), int(i.shape[1])] for i in imgs], np.int32)
    rhw = np.array([resized_size(h, w, resize_to) if resize_to else (h, w) for h, w in hw], np.int32)
    if (rhw[:, 0] > H).any() or (rhw[:, 1] > W).any() or (rhw < 1).any():
        raise ValueError('a resized image does not fit the %dx%d canvas: %s' % (H, W, rhw.tolist()))
    ptrs = (ctypes.c_void_p * B)(*[i.data_ptr() for i in imgs])
    d_lut, d_lut16 = device_luts(mean, std, dev)
    with torch.cuda.device(dev):
        sums = torch.zeros(B, 3, dtype=torch.int64, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if model is not None:
            base, border = model.input_tensor(B, H, W, dev, heads)
            mode, dst, ret = 1, base, None
        else:
            if out is None:
                out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
            mode, dst, border, ret = 0, out.data_ptr(), 0, out
        _lib.check(lib.rtm3d_preprocess_batch(stream, B, ptrs, hw.ctypes.data_as(ctypes.c_void_p), rhw.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_void_p(dst), mode, H, W, border, d_lut.data_ptr(), d_lut16.data_ptr(),
                                              sums.data_ptr()), 'preprocess_batch')
    pads = [((W - int(w)) // 2, (H - int(h)) // 2) for h, w in rhw]
    return ret, pads, [(int(h), int(w)) for h, w in rhw]
