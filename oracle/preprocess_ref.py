"""CPU restatement of the reference's input transform.  TEST INFRASTRUCTURE ONLY.

The reference feeds the attack with ``Compose([Resize(224, bicubic), CenterCrop(224), ToTensor()])`` applied to PIL
images (train/adversarial_training_clip.py:105-116; open_clip's image_processor minus its Normalize).  Both pieces are
third-party and absent from /root/reference: torchvision==0.15.2 (requirements.txt:145) for the geometry, Pillow for the
resampling.  Restated here from their published algorithms:

* ``resize_size`` / ``center_crop_box``: torchvision.transforms.functional.resize (int size = shorter edge, the other
  edge ``int(size * long / short)``) and center_crop (``int(round((h - ch) / 2.0))``);
* ``pil_resize_bicubic_u8``: Pillow's ImagingResample for 8-bit images - separable, antialiased (support scaled by the
  down-scaling factor), Keys bicubic a = -0.5, coefficients normalised in double precision, converted to 22-bit fixed
  point, horizontal pass THEN vertical pass, each rounded and clipped to uint8;
* ``to_tensor``: HWC uint8 -> CHW float32 / 255.

Pinned bit-exactly against Pillow itself (tests/golden/preprocess_*.npz, generated in the build container by
tests/golden/make_golden.py g8).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_size(h: int, w: int, size: int):
    """(new_h, new_w) of torchvision's Resize(size) for an int size."""
    if w <= h:
        return (size, size) if w == h else (int(size * h / w), size)
    return size, int(size * w / h)


def center_crop_box(h: int, w: int, ch: int, cw: int):
    """(top, left) of torchvision's CenterCrop for an image at least as large as the crop."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Per output index: (first input index, tap count) and the fixed-point taps [out_size, ksize] (int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """One separable pass over ``axis`` (0 = vertical, 1 = horizontal) of an [H, W, C] uint8 image."""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.empty(shape, dtype=np.uint8)
    for o in range(n_out):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += np.take(src, x0 + t, axis=axis) * int(kk[o, t])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[o] = v
        else:
            out[:, o] = v
    return out


def pil_resize_bicubic_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Image.resize((new_w, new_h), BICUBIC) of an [H, W, C] uint8 array."""
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        out = _pass(out, *resample_coeffs(w, new_w), axis=1)
    if new_h != h:
        out = _pass(out, *resample_coeffs(h, new_h), axis=0)
    return out


def to_tensor(img_u8: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(img_u8.transpose(2, 0, 1)).astype(np.float32) / np.float32(255))


def preprocess_ref(img_u8: np.ndarray, size: int = 224) -> np.ndarray:
    """Resize(size, bicubic) -> CenterCrop(size) -> ToTensor of one decoded RGB image [H, W, 3] uint8."""
    h, w = img_u8.shape[:2]
    nh, nw = resize_size(h, w, size)
    r = pil_resize_bicubic_u8(img_u8, nh, nw)
    top, left = center_crop_box(nh, nw, size, size)
    return to_tensor(r[top:top + size, left:left + size])
