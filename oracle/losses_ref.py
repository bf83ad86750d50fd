"""CPU oracle: FARE / TeCoA inner losses (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates train/adversarial_training_clip.py:495-528 (``compute_loss``, ``l2``, ``ce``) and the
closure ``ComputeLossWrapper`` (:260-274).  Pinned by tests/golden/losses.npz, which holds outputs
of the reference's own function bodies (AST-extracted by tests/golden/make_golden.py because the
module itself needs torchvision/open_clip/wandb to import).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def l2_ref(out: torch.Tensor, targets: torch.Tensor, reduction: str = "none") -> torch.Tensor:
    """FARE loss, …clip.py:509-521: per-sample sum_d (out-e0)^2, NOT divided by D;
    'mean' -> mean over the batch, anything else -> [B]."""
    if out.shape != targets.shape:
        raise AssertionError(f"{out.shape} != {targets.shape}")
    if out.shape[0] <= 1:
        raise AssertionError("batch size must be > 1")  # …clip.py:513
    per_sample = ((out - targets) ** 2).sum(dim=1)
    return per_sample.mean() if reduction == "mean" else per_sample


def ce_ref(out: torch.Tensor, targets: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """TeCoA loss on logits, …clip.py:523-528."""
    if out.shape[0] != targets.shape[0] or out.shape[0] <= 1:
        raise AssertionError((out.shape, targets.shape))
    return F.cross_entropy(out, targets, reduction=reduction)


def compute_loss_ref(loss_str, embedding, targets, embedding_orig, logit_scale,
                     embedding_text_labels_norm=None, reduction="mean"):
    """…clip.py:495-507.  Note ``logit_scale * T`` is formed first (…clip.py:501)."""
    if loss_str == "l2":
        return l2_ref(embedding, embedding_orig, reduction)
    if loss_str == "ce":
        return ce_ref(embedding @ (logit_scale * embedding_text_labels_norm), targets, reduction)
    raise ValueError(f"loss {loss_str} not supported")


class ComputeLossWrapperRef:
    """…clip.py:260-274: binds e0, T, reduction, loss name, logit_scale=100."""

    def __init__(self, embedding_orig, embedding_text_labels_norm, reduction="mean", loss=None,
                 logit_scale=100.0):
        self.embedding_orig = embedding_orig
        self.embedding_text_labels_norm = embedding_text_labels_norm
        self.reduction = reduction
        self.loss_str = loss
        self.logit_scale = logit_scale

    def __call__(self, embedding, targets):
        return compute_loss_ref(self.loss_str, embedding, targets, self.embedding_orig,
                                self.logit_scale, self.embedding_text_labels_norm, self.reduction)


def compute_acc_ref(logits, targets) -> float:
    """…clip.py:488-492 (argmax ties -> first index)."""
    return (logits.max(dim=1)[1].eq(targets).sum() / targets.shape[0]).item() * 100
