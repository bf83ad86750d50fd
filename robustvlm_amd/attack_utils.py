"""project_perturbation / normalize_grad (vlm_eval/attacks/utils.py:8-26) on device tensors.

The Linf branches are what the fused HIP kernels implement (rvlm_pgd_linf_update); these
tensor-level helpers keep the reference's standalone API (and its L2 branch) available."""
from __future__ import annotations

import torch
import torch.nn.functional as F

LINF = ("inf", "linf", "Linf")
L2 = (2, 2.0, "l2", "L2", "2")


def project_perturbation(perturbation, eps, norm):
    if norm in LINF:
        return torch.clamp(perturbation, -eps, eps)
    if norm in L2:
        return torch.renorm(perturbation, p=2, dim=0, maxnorm=eps)
    raise NotImplementedError(f"Norm {norm} not supported")


def normalize_grad(grad, p):
    if p in LINF:
        return grad.sign()
    if p in L2:
        bs = grad.shape[0]
        return F.normalize(grad.view(bs, -1), p=2, dim=1).view_as(grad)
    raise NotImplementedError(f"Norm {p} not supported")
