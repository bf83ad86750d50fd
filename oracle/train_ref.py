"""CPU oracle: one FARE / TeCoA optimizer step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the part of ``train_one_epoch`` after the attack (train/adversarial_training_clip.py:338-366):
clean / adversarial forward, ``compute_loss``, ``loss_total.backward()``, ``torch.optim.AdamW`` (:196-197)
and the third-party open_clip ``cosine_lr`` (:211).  The reference module cannot be imported here
(torchvision / open_clip / wandb missing), so this file is pinned by construction only: it is plain torch
autograd + torch.optim.AdamW over oracle/vit_ref.py (itself pinned against HF transformers) - "parity
unpinned" by reference outputs for this row.
"""
from __future__ import annotations

import math

import torch

from . import vit_ref as V
from .losses_ref import compute_loss_ref


def cosine_lr_ref(step, base_lr, warmup_length, steps):
    if step < warmup_length:
        return base_lr * (step + 1) / warmup_length
    e, es = step - warmup_length, steps - warmup_length
    return 0.5 * (1 + math.cos(math.pi * e / es)) * base_lr


class TrainStepRef:
    def __init__(self, cfg, weights, lr=1e-5, wd=1e-4, warmup=1400, steps=20000, loss="l2",
                 output_normalize=False, clean_weight=0.0, T=None):
        self.cfg = cfg
        self.w = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
        self.opt = torch.optim.AdamW(list(self.w.values()), lr=lr, weight_decay=wd)
        self.lr, self.warmup, self.steps = lr, warmup, steps
        self.loss, self.on, self.cw, self.T = loss, output_normalize, clean_weight, T
        self.step_total = 0
        self._set_lr(cosine_lr_ref(0, lr, warmup, steps))

    def _set_lr(self, lr):
        for g in self.opt.param_groups:
            g["lr"] = lr

    def forward(self, x):
        e = V.vit_forward(self.cfg, self.w, V.normalize_pixels(x))
        return torch.nn.functional.normalize(e, dim=-1) if self.on else e

    def step(self, x, x_adv, targets, e0):
        loss_clean = 0.0
        if self.cw > 0.:
            loss_clean = compute_loss_ref(self.loss, self.forward(x), targets, e0, 100., None)
        loss = compute_loss_ref(self.loss, self.forward(x_adv), targets, e0, 100., self.T)
        total = self.cw * loss_clean + (1 - self.cw) * loss
        self.opt.zero_grad()
        total.backward()
        grads = {k: v.grad.detach().clone() for k, v in self.w.items()}
        self.opt.step()
        self.step_total += 1
        self._set_lr(cosine_lr_ref(self.step_total, self.lr, self.warmup, self.steps))
        return float(loss.detach()), grads
