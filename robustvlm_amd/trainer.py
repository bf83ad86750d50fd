"""Outer FARE / TeCoA training step on the native engine.

Mirrors the step semantics of ``train_one_epoch`` (train/adversarial_training_clip.py:289-366):

    e0 = model_orig(x)                        (frozen copy, no grad)            :296-297
    x_adv = pgd | apgd (model in eval mode)                                     :305-333
    emb_clean = model(x) ; emb_adv = model(x_adv)      (train mode)             :340,349
    loss_total = cw * loss_clean + (1 - cw) * loss(emb_adv, e0 | targets)       :356-360
    loss_total.backward(); optimizer.step(); zero_grad(); scheduler(step)       :361-366

with AdamW (lr 1e-5, wd 1e-4, :196-197) and open_clip's cosine_lr with linear warm-up (:211).
The reference wraps the model in single-process nn.DataParallel (:184-191); here every rank owns one GPU
and its shard of the batch, the attack needs no communication, and the ONLY collective of a step is one
RCCL all-reduce of the flat fp32 gradient buffer (303.97 M elements for ViT-L/14), followed by the
1/world_size scaling fused into the AdamW kernel.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from . import _lib as L
from .clip_model import ClipVisionModel, ComputeLossWrapper, compute_loss
from .config import state_dict_shapes
from .engine import VitEngine
from .pgd_train import pgd
from .apgd_train import apgd_train


def cosine_lr_value(step: int, base_lr: float, warmup_length: int, steps: int) -> float:
    """open_clip ``training.scheduler.cosine_lr`` (third party, …clip.py:18,211): linear warm-up
    ``lr*(step+1)/warmup`` then ``0.5*(1+cos(pi*e/es))*lr``."""
    if step < warmup_length:
        return base_lr * (step + 1) / warmup_length
    e, es = step - warmup_length, steps - warmup_length
    return 0.5 * (1 + math.cos(math.pi * e / es)) * base_lr


class FlatParams:
    """All parameters of the vision tower as ONE flat fp32 buffer (+ per-key views), so that the gradient
    all-reduce and the optimizer are single launches over 304 M elements."""

    def __init__(self, cfg, state_dict: dict, device):
        self.shapes = state_dict_shapes(cfg)
        self.offsets, n = {}, 0
        for k, shp in self.shapes.items():
            cnt = 1
            for d in shp:
                cnt *= d
            self.offsets[k] = (n, cnt)
            n = (n + cnt + 3) // 4 * 4        # keep every tensor 16-byte aligned
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + c].view(self.shapes[k]) for k, (o, c) in self.offsets.items()}
        if state_dict is not None:
            for k, v in self.views.items():
                v.copy_(state_dict[k].detach().to(device=device, dtype=torch.float32))

    def like(self):
        other = FlatParams.__new__(FlatParams)
        other.shapes, other.offsets, other.numel = self.shapes, self.offsets, self.numel
        other.flat = torch.zeros_like(self.flat)
        other.views = {k: other.flat[o:o + c].view(self.shapes[k]) for k, (o, c) in self.offsets.items()}
        return other

    def state_dict(self):
        return {k: v.clone() for k, v in self.views.items()}


class AdversarialTrainer:
    """One process per GPU.  ``state_dict`` = open_clip ``visual.state_dict()`` of the model to fine-tune;
    the frozen ``model_orig`` copy is created from the same weights (…clip.py:95-97,183-186)."""

    def __init__(self, cfg, state_dict, batch_size, precision="bf16", lr=1e-5, wd=1e-4, warmup=1400,
                 steps=20000, loss="l2", inner_loss="l2", attack="pgd", norm="linf", eps=4 / 255,
                 iterations_adv=10, stepsize_adv=1 / 255, output_normalize=False, clean_weight=0.0,
                 embedding_text_labels_norm=None, betas=(0.9, 0.999), adam_eps=1e-8, device=None):
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.cfg = cfg
        self.lib = L.load()
        self.params = FlatParams(cfg, state_dict, self.device)
        self.grads = self.params.like()
        self.exp_avg = torch.zeros_like(self.params.flat)
        self.exp_avg_sq = torch.zeros_like(self.params.flat)
        self.engine = VitEngine(cfg, self.params.views, precision=precision, max_batch=batch_size,
                                device=self.device, trainable=True)
        self.engine_orig = VitEngine(cfg, self.params.views, precision=precision, max_batch=batch_size,
                                     device=self.device, inference_only=True)
        self.model = ClipVisionModel(self.engine)
        self.model_orig = ClipVisionModel(self.engine_orig).eval()
        self.lr, self.wd, self.warmup, self.steps = lr, wd, warmup, steps
        self.betas, self.adam_eps = betas, adam_eps
        self.loss, self.inner_loss, self.attack, self.norm = loss, inner_loss, attack, norm
        self.eps, self.iterations_adv, self.stepsize_adv = eps, iterations_adv, stepsize_adv
        self.output_normalize, self.clean_weight = output_normalize, clean_weight
        self.T = embedding_text_labels_norm
        self.step_total = 0
        self.cur_lr = cosine_lr_value(0, lr, warmup, steps)      # scheduler(start_step), …clip.py:219
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    # -- pieces of the step ----------------------------------------------------------------------
    def _attack(self, data, targets, e0):
        wrap = ComputeLossWrapper(e0, self.T, "none" if self.attack == "apgd" else "mean", self.inner_loss, 100.)
        self.model.eval()                                                       # …clip.py:305
        if self.attack == "pgd":
            adv = pgd(self.model, wrap, data, targets, self.norm, self.eps, self.iterations_adv, self.stepsize_adv,
                      self.output_normalize,
                      perturbation=torch.zeros_like(data).uniform_(-self.eps, self.eps), mode="max")
        elif self.attack == "apgd":
            adv = apgd_train(self.model, data, targets, self.norm, self.eps, n_iter=self.iterations_adv, loss_fn=wrap)
        elif self.attack == "none":
            adv = data
        else:
            raise ValueError(f"attack {self.attack} not supported")
        self.model.train()                                                      # …clip.py:338
        return adv

    def _loss_backward(self, x, targets, e0, weight, accumulate):
        """forward (activations kept for wgrad) + loss + weight gradients scaled by ``weight``."""
        emb = self.engine.forward(x, None, self.output_normalize, save=2)
        e = emb.detach().requires_grad_(True)
        loss = compute_loss(self.loss, e, targets, e0, 100., self.T)            # reduction='mean'
        (d_emb,) = torch.autograd.grad(loss, e)
        self.engine.backward_params(d_emb * weight, self.grads.views, accumulate=accumulate)
        return loss.detach(), emb

    def train_step(self, data, targets, data_adv=None):
        """One optimizer step on this rank's shard; returns dict(loss, loss_clean, lr).
        ``data_adv`` (optional) bypasses the attack with precomputed adversarial images (tests)."""
        with torch.no_grad():
            e0 = self.model_orig(data, self.output_normalize)                   # …clip.py:296-297
        if data_adv is None:
            data_adv = self._attack(data, targets, e0)
        cw = self.clean_weight
        loss_clean = torch.zeros((), device=self.device)
        accumulate = False
        if cw > 0.:                                                              # …clip.py:341-347
            loss_clean, _ = self._loss_backward(data, targets, e0, cw, False)
            accumulate = True
        loss, _ = self._loss_backward(data_adv, targets, e0, 1.0 - cw, accumulate)   # …clip.py:349-361
        if self.world > 1:
            dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM)              # the step's only collective
        self.step_total += 1
        b1, b2 = self.betas
        with torch.cuda.device(self.device):
            L.check(self.lib.rvlm_adamw_step(self.params.flat.data_ptr(), self.grads.flat.data_ptr(),
                                             self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.params.numel,
                                             float(self.cur_lr), b1, b2, self.adam_eps, self.wd, self.step_total,
                                             1.0 / self.world, L.stream_ptr()), "rvlm_adamw_step")
        self.engine.load_state_dict(self.params.views)       # refresh the bf16 / transposed GEMM copies
        lr_used = self.cur_lr
        self.cur_lr = cosine_lr_value(self.step_total, self.lr, self.warmup, self.steps)   # scheduler(step_total)
        return dict(loss=loss, loss_clean=loss_clean, lr=lr_used)

    def state_dict(self):
        """open_clip ``visual.state_dict()`` of the fine-tuned tower (what …clip.py:239,470 saves)."""
        return self.params.state_dict()

    def optimizer_state_dict(self):
        """AdamW state keyed by parameter name (see robustvlm_amd/checkpoint.py)."""
        def split(flat):
            return {k: flat[o:o + c].view(self.params.shapes[k]).clone() for k, (o, c) in self.params.offsets.items()}
        return {"step": self.step_total, "exp_avg": split(self.exp_avg), "exp_avg_sq": split(self.exp_avg_sq),
                "lr": self.lr, "wd": self.wd, "betas": self.betas, "eps": self.adam_eps}

    def load_optimizer_state_dict(self, sd, start_step=None):
        """Resume (…clip.py:207-208,219): restores the moments and the step counter; the LR schedule is a pure
        function of the step."""
        for k, (o, c) in self.params.offsets.items():
            self.exp_avg[o:o + c].copy_(sd["exp_avg"][k].reshape(-1).to(self.device))
            self.exp_avg_sq[o:o + c].copy_(sd["exp_avg_sq"][k].reshape(-1).to(self.device))
        self.step_total = int(sd["step"] if start_step is None else start_step)
        self.cur_lr = cosine_lr_value(self.step_total, self.lr, self.warmup, self.steps)
