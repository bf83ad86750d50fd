"""CPU oracle: one FARE / TeCoA optimizer step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the part of ``train_one_epoch`` after the attack (train/adversarial_training_clip.py:338-387):
clean / adversarial forward, ``compute_loss`` (``--loss_clean`` with T=None for the clean term, ``--trades``),
``loss_total.backward()``, ``torch.optim.AdamW`` (:196-197), the third-party open_clip ``cosine_lr`` (:211) and the
logging metrics cos-sim-clean / cos-sim / acc / racc (:368-387); ``eval_ref`` restates the periodic validation
(:389-424).

PINNED: the reference module cannot be imported here (torchvision / open_clip / wandb missing), but its
``train_one_epoch`` takes model, optimizer, scheduler, dataloader and ``args`` as parameters, so
tests/golden/make_golden_train.py AST-extracts it and RUNS it on a tiny seeded ViT (4 recipes x 5 steps);
tests/test_oracle_golden.py::test_train_step_oracle_vs_reference_train_one_epoch holds this file to the recorded
losses, metrics, learning rates and post-step parameters bit for bit.  What the fixture caught: the reference never
calls ``scheduler`` before the first optimizer step (:211-219 build it, :366 is its only call site, AFTER
``step_total += 1``), so step 1 runs at the BASE learning rate and step s >= 2 at ``cosine_lr(s - 1)``.
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
                 output_normalize=False, clean_weight=0.0, T=None, loss_clean="l2", trades=False):
        self.cfg = cfg
        self.w = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
        self.opt = torch.optim.AdamW(list(self.w.values()), lr=lr, weight_decay=wd)
        self.lr, self.warmup, self.steps = lr, warmup, steps
        self.loss, self.on, self.cw, self.T = loss, output_normalize, clean_weight, T
        self.loss_clean, self.trades = loss_clean, trades
        self.last_metrics = {}
        self.step_total = 0
        # no scheduler call before the first step (…clip.py:211-219): AdamW starts at the base LR it was built with

    def _set_lr(self, lr):
        for g in self.opt.param_groups:
            g["lr"] = lr

    def forward(self, x):
        e = V.vit_forward(self.cfg, self.w, V.normalize_pixels(x))
        return torch.nn.functional.normalize(e, dim=-1) if self.on else e

    def step(self, x, x_adv, targets, e0):
        emb_clean = self.forward(x)                                                    # :340
        loss_clean = 0.0
        if self.cw > 0.:                                                               # :341-347
            loss_clean = compute_loss_ref(self.loss_clean, emb_clean, targets, e0, 100., None)
        emb_adv = self.forward(x_adv)                                                  # :349
        e_ref = emb_clean.detach().clone() if self.trades else e0                      # :352-358
        loss = compute_loss_ref(self.loss, emb_adv, targets, e_ref, 100., self.T)
        total = self.cw * loss_clean + (1 - self.cw) * loss                            # :360
        self.opt.zero_grad()
        total.backward()
        grads = {k: v.grad.detach().clone() for k, v in self.w.items()}
        self.opt.step()
        self.step_total += 1
        self._set_lr(cosine_lr_ref(self.step_total, self.lr, self.warmup, self.steps))
        with torch.no_grad():                                                          # :368-387
            F = torch.nn.functional
            m = {"cos_sim_clean": float(F.cosine_similarity(emb_clean, e0, dim=1).mean()),
                 "cos_sim": float(F.cosine_similarity(emb_adv, e0, dim=1).mean()),
                 "loss_total": float(total), "loss_clean": float(loss_clean)}
            if isinstance(targets, torch.Tensor) and self.T is not None:
                acc = lambda lg: (lg.max(dim=1)[1].eq(targets).sum() / targets.shape[0]).item() * 100   # noqa: E731
                m["racc"] = acc(emb_adv @ self.T)
                m["acc"] = acc(F.normalize(emb_clean, dim=1) @ self.T)
            self.last_metrics = m
        return float(loss.detach()), grads


def eval_ref(cfg, weights, x_eval, y_eval, T, eps, clean_weight, norm="linf"):
    """The periodic validation of train_one_epoch (…clip.py:389-424): 50-step supervised APGD (CE on the zero-shot
    head; ``initial_stepsize = 0.05 * eps`` iff clean_weight > 0), then acc / racc of the NORMALISED embeddings and the
    clean-vs-adversarial cosine similarity.  Returns (acc, racc, cos_sim, x_adv)."""
    from . import attacks_ref as A
    from .losses_ref import ComputeLossWrapperRef, compute_acc_ref
    model = V.ClipVisionModelRef(cfg, weights).eval()
    wrap = ComputeLossWrapperRef(None, T, "none", "ce", 100.)
    adv = A.apgd_train_ref(model, x_eval, y_eval, norm, eps, n_iter=50, loss_fn=wrap,
                           initial_stepsize=0.05 * eps if clean_weight > 0 else None)
    with torch.no_grad():
        ea, ec = model(adv, True), model(x_eval, True)
        racc, acc = compute_acc_ref(ea @ T, y_eval), compute_acc_ref(ec @ T, y_eval)
        cs = float(torch.nn.functional.cosine_similarity(ea, ec, dim=1).mean())
    return acc, racc, cs, adv
