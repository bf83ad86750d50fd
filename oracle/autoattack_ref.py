"""CPU restatement of the AutoAttack orchestration the reference runs for config 5.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and nothing else.

Follows, line by line:
  * ``APGDAttack_targeted.perturb``                       autoattack/autopgd_base.py:621-707
  * ``AutoAttack.run_standard_evaluation`` (robust-flag bookkeeping, per-attack loop over the still-robust points,
    ``bs`` batching, seeds)                                autoattack/autoattack.py:80-241
for the attacks CLIP_eval/clip_robustbench.py:149 selects (``apgd-ce``, ``apgd-t``) plus ``apgd-dlr``.
Pinned bit-exactly by tests/golden/autoattack_tiny.npz and dlr_losses.npz (outputs of the reference itself).
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .attacks_ref import APGDAttackRef


class APGDAttackTargetedRef(APGDAttackRef):
    def __init__(self, predict, n_iter=100, norm="Linf", n_restarts=1, eps=None, seed=0, eot_iter=1, rho=.75,
                 topk=None, n_target_classes=9, verbose=False, device=None, use_largereps=False, is_tf_model=False,
                 logger=None, alpha=None, use_rs=True):
        super().__init__(predict, n_iter=n_iter, norm=norm, n_restarts=n_restarts, eps=eps, seed=seed,
                         loss="dlr-targeted", eot_iter=eot_iter, rho=rho, alpha=alpha, use_rs=use_rs)
        self.n_target_classes = n_target_classes

    def perturb(self, x, y=None, x_init=None):
        x = x.detach().clone().float()
        y_pred = self.model(x).max(1)[1]                                          # :634
        y = y_pred.detach().clone().long() if y is None else y.detach().clone().long()
        adv = x.clone()
        acc = y_pred == y
        torch.random.manual_seed(self.seed)                                       # :655
        for target_class in range(2, self.n_target_classes + 2):                  # :672
            for _ in range(self.n_restarts):
                ind = acc.nonzero().squeeze(1)
                if ind.numel() == 0:
                    continue
                x_to_fool, y_to_fool = x[ind].clone(), y[ind].clone()
                output = self.model(x_to_fool)
                self.y_target = output.sort(dim=1)[1][:, -target_class]          # :686
                _, acc_curr, _, adv_curr = self.attack_single_run(x_to_fool, y_to_fool)
                fooled = (acc_curr == 0).nonzero().squeeze(1)
                acc[ind[fooled]] = False
                adv[ind[fooled]] = adv_curr[fooled].clone()
        return adv


class AutoAttackRef:
    def __init__(self, model, norm="Linf", eps=.3, seed=None, verbose=False, attacks_to_run=(), version="custom",
                 device="cpu", alpha=None, iterations_apgd=100, use_rs=True):
        assert norm in ("Linf", "L2") and version == "custom"   # (L2: the two APGD stages; the square stage is restated for Linf)
        self.model, self.epsilon, self.seed = model, eps, seed
        self.attacks_to_run = list(attacks_to_run)
        self.apgd = APGDAttackRef(model, n_restarts=5, n_iter=iterations_apgd, eps=eps, norm=norm, eot_iter=1, rho=.75,
                                  seed=seed, alpha=alpha, use_rs=use_rs)                      # autoattack.py:34-36
        self.apgd_targeted = APGDAttackTargetedRef(model, n_restarts=1, n_iter=iterations_apgd, eps=eps, norm=norm,
                                                   eot_iter=1, rho=.75, seed=seed, alpha=alpha, use_rs=use_rs)  # :47-49
        from .square_ref import SquareAttackRef
        self.square = SquareAttackRef(model, p_init=.8, n_queries=5000, eps=eps, norm=norm, n_restarts=1, seed=seed,
                                      resc_schedule=False) if norm == "Linf" else None        # :42-44

    def get_seed(self):
        return time.time() if self.seed is None else self.seed                    # :79-80

    def _checks(self, x, y):
        """autoattack.py:113-120 -> checks.py: 5 + 1 + 1 model calls on the first batch; warnings only."""
        import warnings
        with torch.no_grad():
            outs, corr = [], []
            for _ in range(5):                                                    # check_randomized
                o = self.model(x)
                corr.append((o.max(1)[1] == y).sum().item())
                outs.append(o / ((o ** 2).reshape(o.shape[0], -1).sum(-1, keepdim=True).sqrt() + 1e-10))
            max_diff = max(float(((outs[c] - outs[e]) ** 2).reshape(x.shape[0], -1).sum(-1).sqrt().max())
                           for c in range(4) for e in range(c + 1, 5))
            if any(c != corr[-1] for c in corr) or max_diff > 1e-4:
                warnings.warn("it seems to be a randomized defense!")
            o = self.model(x)                                                     # check_range_output
            n_cls = o.shape[-1]
        self.model(x)                                                             # check_dynamic (traced call)
        targeted = "apgd-t" in self.attacks_to_run
        if ("apgd-dlr" in self.attacks_to_run or targeted) and (n_cls <= 3 or (
                targeted and self.apgd_targeted.n_target_classes + 1 > n_cls)):    # check_n_classes
            warnings.warn("too few classes for the (targeted) DLR loss")
        return n_cls

    def run_standard_evaluation(self, x_orig, y_orig, bs=250, return_labels=False):
        self._checks(x_orig[:bs], y_orig[:bs])
        with torch.no_grad():
            n_batches = int(np.ceil(x_orig.shape[0] / bs))                        # :124-139
            robust_flags = torch.zeros(x_orig.shape[0], dtype=torch.bool)
            y_adv = torch.empty_like(y_orig)
            for b in range(n_batches):
                s, e = b * bs, min((b + 1) * bs, x_orig.shape[0])
                output = self.model(x_orig[s:e].clone()).max(dim=1)[1]
                y_adv[s:e] = output
                robust_flags[s:e] = y_orig[s:e].eq(output)
            x_adv = x_orig.clone().detach()
            for attack in self.attacks_to_run:                                    # :157
                num_robust = torch.sum(robust_flags).item()
                if num_robust == 0:
                    break
                n_batches = int(np.ceil(num_robust / bs))
                robust_lin_idcs = torch.nonzero(robust_flags, as_tuple=False).squeeze(1)
                for b in range(n_batches):
                    idcs = robust_lin_idcs[b * bs:min((b + 1) * bs, num_robust)]
                    x, y = x_orig[idcs].clone(), y_orig[idcs].clone()
                    if attack == "apgd-ce":                                       # :184-188
                        self.apgd.loss, self.apgd.seed = "ce", self.get_seed()
                        with torch.enable_grad():
                            adv_curr = self.apgd.perturb(x, y)
                    elif attack == "apgd-dlr":
                        self.apgd.loss, self.apgd.seed = "dlr", self.get_seed()
                        with torch.enable_grad():
                            adv_curr = self.apgd.perturb(x, y)
                    elif attack == "apgd-t":                                      # :207-210
                        self.apgd_targeted.seed = self.get_seed()
                        with torch.enable_grad():
                            adv_curr = self.apgd_targeted.perturb(x, y)
                    elif attack == "square":                                      # :202-205
                        self.square.seed = self.get_seed()
                        adv_curr = self.square.perturb(x, y)
                    else:
                        raise ValueError("Attack not supported")
                    output = self.model(adv_curr).max(dim=1)[1]                   # :221-229
                    false_batch = ~y.eq(output)
                    non_robust = idcs[false_batch]
                    robust_flags[non_robust] = False
                    x_adv[non_robust] = adv_curr[false_batch].detach()
                    y_adv[non_robust] = output[false_batch].detach()
        return (x_adv, y_adv) if return_labels else x_adv
