"""Input front end on the device.

``ResizeCenterCropToTensor(224)`` is the transform the reference runs on its DataLoader workers -
``Compose([Resize(224, bicubic), CenterCrop(224), ToTensor()])`` (train/adversarial_training_clip.py:105-116, the
open_clip image processor minus Normalize) - applied to DECODED images (uint8 HWC tensors on the GPU; JPEG decoding is
not part of this path).  Outputs are bit-identical to torchvision-over-Pillow: see csrc/preprocess.hip.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class ResizeCenterCropToTensor:
    def __init__(self, size: int = 224, max_input_dim: int = 8192):
        self.size = int(size)
        self.lib = L.load()
        self._h = C.c_void_p()
        L.check(self.lib.rvlm_preproc_create(self.size, int(max_input_dim), C.byref(self._h)), "rvlm_preproc_create")

    def __call__(self, img: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """img: uint8 [H, W, 3] CUDA tensor (a decoded RGB image) -> float32 [3, size, size] in [0, 1]."""
        if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3
                and img.shape[2] == 3):
            raise ValueError("expected a uint8 CUDA tensor of shape [H, W, 3]")
        img = img.contiguous()
        if out is None:
            out = torch.empty(3, self.size, self.size, dtype=torch.float32, device=img.device)
        elif not (isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.device == img.device
                  and out.is_contiguous() and tuple(out.shape) == (3, self.size, self.size)):
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(3, self.size, self.size)} on {img.device}")
        with torch.cuda.device(img.device):
            L.check(self.lib.rvlm_preproc_run(self._h, img.data_ptr(), img.shape[0], img.shape[1], out.data_ptr(),
                                              L.stream_ptr()), "rvlm_preproc_run")
        return out

    def batch(self, images, out: torch.Tensor | None = None) -> torch.Tensor:
        """List of decoded images (uint8 [H, W, 3] CUDA tensors of any mix of shapes) -> [B, 3, size, size] in ONE kernel
        launch (rvlm_preproc_run_batch): what the reference's DataLoader collates from its 8 workers
        (train/adversarial_training_clip.py:119-148)."""
        n = len(images)
        if n == 0:
            raise ValueError("empty batch")
        keep, dev = [], None
        for im in images:
            if not (isinstance(im, torch.Tensor) and im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3
                    and im.shape[2] == 3 and (dev is None or im.device == dev)):
                raise ValueError("expected uint8 CUDA tensors of shape [H, W, 3] on one device")
            dev = im.device
            keep.append(im.contiguous())
        if out is None:
            out = torch.empty(n, 3, self.size, self.size, dtype=torch.float32, device=dev)
        elif not (isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.device == dev and out.is_contiguous()
                  and tuple(out.shape) == (n, 3, self.size, self.size)):
            # the kernel writes n * 3 * size * size floats through a raw pointer: anything else would be out of bounds
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(n, 3, self.size, self.size)} on {dev}")
        ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in keep])
        hs = (C.c_int * n)(*[im.shape[0] for im in keep])
        ws = (C.c_int * n)(*[im.shape[1] for im in keep])
        with torch.cuda.device(dev):
            L.check(self.lib.rvlm_preproc_run_batch(self._h, ptrs, hs, ws, n, out.data_ptr(), L.stream_ptr()),
                    "rvlm_preproc_run_batch")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rvlm_preproc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
