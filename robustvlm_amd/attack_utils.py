"""project_perturbation / normalize_grad (vlm_eval/attacks/utils.py:8-26) on device tensors, as kernels of librvlm
(rvlm_project_perturbation, rvlm_normalize_grad).  Inside pgd() the same arithmetic is fused into the per-iteration update
kernels (rvlm_pgd_linf_update / rvlm_pgd_l2_update); these keep the reference's standalone API.

Restrictions against the reference's plain torch ops (documented, deliberate): tensors must live on the GPU (no CPU route:
the product path never computes on the host), the arithmetic is fp32 (half inputs are widened, the result is cast back
to the input's dtype), and the result is detached (the reference's callers only ever use these under no_grad)."""
from __future__ import annotations

import torch

from . import _lib as L
from .engine import _require_cuda, _f32c

LINF = ("inf", "linf", "Linf")
L2 = (2, 2.0, "l2", "L2", "2")


def _norm_kind(norm):
    if norm in LINF:
        return 0
    if norm in L2:
        return 2
    raise NotImplementedError(f"Norm {norm} not supported")               # utils.py:16


def project_perturbation(perturbation, eps, norm):
    kind = _norm_kind(norm)
    _require_cuda(perturbation, "perturbation")
    p = _f32c(perturbation)
    out = torch.empty_like(p)
    with torch.cuda.device(p.device):
        L.check(L.load().rvlm_project_perturbation(p.data_ptr(), p[0].numel(), p.shape[0], kind, float(eps), out.data_ptr(),
                                                   L.stream_ptr()), "rvlm_project_perturbation")
    return out if out.dtype == perturbation.dtype else out.to(perturbation.dtype)


def normalize_grad(grad, p):
    kind = _norm_kind(p)
    _require_cuda(grad, "grad")
    g = _f32c(grad)
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        L.check(L.load().rvlm_normalize_grad(g.data_ptr(), g[0].numel(), g.shape[0], kind, out.data_ptr(), L.stream_ptr()),
                "rvlm_normalize_grad")
    return out if out.dtype == grad.dtype else out.to(grad.dtype)
