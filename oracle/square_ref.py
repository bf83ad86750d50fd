"""CPU restatement of the L-inf Square Attack (Andriushchenko et al., arXiv:1912.00049) as the reference runs it for the
black-box evaluation (CLIP_eval/clip_robustbench.py:150-151 -> AutoAttack(version='custom', attacks_to_run=['square'])).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and nothing else.

Follows autoattack/square.py:
  * margin / loss of a batch of logits                      :68-86
  * the p schedule                                          :192-219
  * the L-inf single run (vertical-stripe start, one random square per query shared by the whole batch, accept when
    the loss improves or the point becomes misclassified)   :221-300
  * perturb(): restarts over the still-correct points       :549-618
The random numbers are drawn with torch.rand on the CPU generator in exactly the reference's order (the reference
draws on the CPU and moves the result to the device, :113-119), so a seed reproduces its trajectory bit for bit.
Pinned by tests/golden/square_tiny.npz (outputs of the reference itself).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def p_schedule(it: int, n_queries: int, p_init: float, rescale: bool) -> float:
    """Fraction of pixels a square covers at query ``it`` (square.py:192-219): halved at fixed query counts."""
    if rescale:
        it = int(it / n_queries * 10000)
    for bound, div in ((10, 1), (50, 2), (200, 4), (500, 8), (1000, 16), (2000, 32), (4000, 64), (6000, 128),
                       (8000, 256)):
        if it <= bound:
            return p_init / div
    return p_init / 512


class SquareAttackRef:
    def __init__(self, predict, norm="Linf", n_queries=5000, eps=None, p_init=.8, n_restarts=1, seed=0,
                 targeted=False, loss="margin", resc_schedule=True):
        assert norm == "Linf" and eps is not None and loss in ("ce", "margin")
        self.predict, self.n_queries, self.eps, self.p_init = predict, n_queries, eps, p_init
        self.n_restarts, self.seed, self.targeted, self.loss, self.rescale = n_restarts, seed, targeted, loss, resc_schedule
        self.queries_used = None

    # square.py:113-119
    @staticmethod
    def _signs(shape):
        return torch.sign(2 * torch.rand(shape) - 1)

    @staticmethod
    def _randint(low, high):
        return (low + (high - low) * torch.rand([1])).long()

    def margin_and_loss(self, x, y):
        logits = self.predict(x)
        xent = F.cross_entropy(logits, y, reduction="none")
        rows = torch.arange(x.shape[0])
        z_y = logits[rows, y].clone()
        logits[rows, y] = -float("inf")
        z_other = logits.max(dim=-1)[0]
        if self.targeted:
            return z_other - z_y, xent
        return (z_y - z_other, -1. * xent) if self.loss == "ce" else (z_y - z_other, z_y - z_other)

    def attack_single_run(self, x, y):
        with torch.no_grad():
            c, h, w = x.shape[1:]
            n_features = c * h * w
            x_best = torch.clamp(x + self.eps * self._signs([x.shape[0], c, 1, w]), 0., 1.)      # :235-236
            margin_min, loss_min = self.margin_and_loss(x_best, y)
            n_queries = torch.ones(x.shape[0])
            if (margin_min < 0.0).all():
                return n_queries, x_best
            for it in range(self.n_queries):
                todo = (margin_min > 0.0).nonzero().flatten()
                x_c, xb_c, y_c = x[todo], x_best[todo], y[todo]
                p = p_schedule(it, self.n_queries, self.p_init, self.rescale)
                s = min(max(int(round(math.sqrt(p * n_features / c))), 1), min(h, w))
                vh = int(self._randint(0, h - s))
                vw = int(self._randint(0, w - s))
                window = torch.zeros([c, h, w])
                window[:, vh:vh + s, vw:vw + s] = 2. * self.eps * self._signs([c, 1, 1])
                x_new = torch.clamp(torch.min(torch.max(xb_c + window, x_c - self.eps), x_c + self.eps), 0., 1.)
                margin, loss = self.margin_and_loss(x_new, y_c)
                better = (loss < loss_min[todo]).float()
                loss_min[todo] = better * loss + (1. - better) * loss_min[todo]
                take = torch.max(better, (margin <= 0.).float())                                 # :284-285
                margin_min[todo] = take * margin + (1. - take) * margin_min[todo]
                t4 = take.reshape(-1, 1, 1, 1)
                x_best[todo] = t4 * x_new + (1. - t4) * xb_c
                n_queries[todo] += 1.
                if (margin_min <= 0.).all():
                    break
            return n_queries, x_best

    def perturb(self, x, y=None):
        adv = x.clone()
        if y is None:
            assert not self.targeted, "the oracle covers the untargeted / given-target calls"
            with torch.no_grad():
                y = self.predict(x).max(1)[1].detach().clone().long()
        else:
            y = y.detach().clone().long()
        pred = self.predict(x).max(1)[1]
        acc = (pred != y) if self.targeted else (pred == y)
        torch.random.manual_seed(self.seed)
        for _ in range(self.n_restarts):
            todo = acc.nonzero().flatten()
            if todo.numel() == 0:
                continue
            nq, cand = self.attack_single_run(x[todo].clone(), y[todo].clone())
            self.queries_used = nq
            out = self.predict(cand).max(1)[1]
            still = (out != y[todo]) if self.targeted else (out == y[todo])
            broken = (still == 0).nonzero().flatten()
            acc[todo[broken]] = 0
            adv[todo[broken]] = cand[broken].clone()
        return adv
