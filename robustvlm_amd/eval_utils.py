"""Evaluation helpers around the zero-shot classifier of BASELINE config 5.

* :func:`compute_accuracy_no_dataloader` - CLIP_eval/eval_utils.py:88-112
* :func:`zeroshot_head`                  - the prompt-ensembling arithmetic of CLIP_eval/clip_robustbench.py:211-224
  (the text tower itself is out of scope: pass the per-template text embeddings)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


@torch.inference_mode()
def compute_accuracy_no_dataloader(model, data, targets, device, batch_size=1000):
    """Top-1 accuracy of ``model`` on tensors ``data`` / ``targets`` in chunks of ``batch_size``; the model's
    training flag is restored afterwards."""
    was_training = getattr(model, "training", False)
    if hasattr(model, "eval"):
        model.eval()
    n_total = n_correct = 0
    for s in range(0, data.shape[0], batch_size):
        xb = data[s:s + batch_size].clone().to(device)
        yb = targets[s:s + batch_size].clone().to(device)
        preds = F.softmax(model(xb), dim=1).max(dim=1)[1]
        n_total += yb.size(0)
        n_correct += preds.eq(yb).sum().item()
    if was_training:
        model.train()
    return n_correct / n_total


def zeroshot_head(template_embeddings) -> torch.Tensor:
    """Column-normalised zero-shot head ``T`` [D, C] from text embeddings.

    ``template_embeddings``: tensor [C, n_templates, D] (or a list of C tensors [n_templates_c, D]) of UN-normalised
    text-tower outputs, one row per prompt template.  Per class: normalise every template embedding, average, normalise
    again (clip_robustbench.py:219-222); a single template ([C, 1, D] or [C, D]) gives the 'std' head (:192-209)."""
    cols = []
    if isinstance(template_embeddings, torch.Tensor) and template_embeddings.dim() == 2:
        template_embeddings = template_embeddings.unsqueeze(1)
    for emb in template_embeddings:
        c = F.normalize(emb.float(), dim=-1).mean(dim=0)
        cols.append(c / c.norm())
    T = torch.stack(cols, dim=1)
    assert torch.allclose(F.normalize(T, dim=0), T, atol=1e-6)
    return T
