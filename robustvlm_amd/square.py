"""Drop-in ``SquareAttack`` (autoattack/square.py) for the black-box evaluation route
(CLIP_eval/clip_robustbench.py:150-151 -> ``AutoAttack(version='custom', attacks_to_run=['square'])``): L-inf, margin /
cross-entropy loss, untargeted or targeted, random restarts over the still-correct points.

A query-only attack: the device work is the model forward (``predict``, e.g. :class:`ClassificationModel` over the
engine) plus two HIP kernels per query - ``rvlm_square_linf_propose`` builds the candidates of the still-robust images,
``rvlm_square_accept`` keeps the ones that improved.  The random numbers are drawn on the CPU generator in the
reference's order (square.py:113-119 draws with ``torch.rand(shape)`` and moves the result to the device), so a seed
reproduces the reference's sequence of squares.  L2 / L1 raise NotImplementedError (not selected by the repo's configs).
"""
from __future__ import annotations

import math
import time

import torch
import torch.nn.functional as F

from . import _lib as L
from .engine import _require_cuda, _f32c


def p_selection(it, n_queries, p_init, rescale):
    """Fraction of the image one square covers at query ``it`` (square.py:192-219)."""
    if rescale:
        it = int(it / n_queries * 10000)
    halvings = sum(it > b for b in (10, 50, 200, 500, 1000, 2000, 4000, 6000, 8000))
    return p_init / (1 << halvings)


class SquareAttack():
    """Square Attack https://arxiv.org/abs/1912.00049 (constructor of square.py:36-66)."""

    def __init__(self, predict, norm='Linf', n_queries=5000, eps=None, p_init=.8, n_restarts=1, seed=0, verbose=False,
                 targeted=False, loss='margin', resc_schedule=True, device=None):
        self.predict = predict
        self.norm = norm
        self.n_queries = n_queries
        self.eps = eps
        self.p_init = p_init
        self.n_restarts = n_restarts
        self.seed = seed
        self.verbose = verbose
        self.targeted = targeted
        self.loss = loss
        self.rescale_schedule = resc_schedule
        self.device = device
        self.return_all = False
        if norm != 'Linf':
            raise NotImplementedError("native SquareAttack covers norm='Linf'")

    def margin_and_loss(self, x, y):
        """:param y: correct labels if untargeted else target labels  (square.py:68-86)"""
        with torch.no_grad():
            logits = self.predict(x).float().clone()
        xent = F.cross_entropy(logits, y, reduction='none')   # bit-identical with the reference's bookkeeping
        z_y = logits.gather(1, y.view(-1, 1)).squeeze(1)
        z_other = logits.scatter(1, y.view(-1, 1), -float('inf')).max(dim=-1)[0]
        if self.targeted:
            return z_other - z_y, xent
        if self.loss == 'ce':
            return z_y - z_other, -1. * xent
        return z_y - z_other, z_y - z_other

    def init_hyperparam(self, x):
        assert self.norm in ['Linf', 'L2', 'L1']
        assert not self.eps is None
        assert self.loss in ['ce', 'margin']
        if self.device is None:
            self.device = x.device
        self.orig_dim = list(x.shape[1:])
        self.ndims = len(self.orig_dim)
        if self.seed is None:
            self.seed = time.time()

    # the reference's two random primitives (CPU generator, then to the device)
    def random_choice(self, shape):
        return torch.sign(2 * torch.rand(shape) - 1).to(self.device)

    def random_int(self, low=0, high=1, shape=[1]):
        return (low + (high - low) * torch.rand(shape)).long().to(self.device)

    def random_target_classes(self, y_pred, n_classes):
        y = torch.zeros_like(y_pred)
        for i in range(y_pred.shape[0]):
            others = [k for k in range(n_classes) if k != int(y_pred[i])]
            y[i] = others[int(self.random_int(0, len(others)))]
        return y.long().to(self.device)

    def attack_single_run(self, x, y):
        """Returns (queries used per sample, x_best) like square.py:221-300."""
        lib = L.load()
        x = _f32c(x)
        n, c, h, w = x.shape
        n_features = c * h * w
        eps = float(self.eps)
        with torch.no_grad(), torch.cuda.device(x.device):
            x_best = torch.clamp(x + eps * self.random_choice([n, c, 1, w]), 0., 1.).contiguous()
            margin_min, loss_min = self.margin_and_loss(x_best, y)
            n_queries = torch.ones(n, device=x.device)
            if (margin_min < 0.0).all():
                return n_queries, x_best
            x_new_buf = torch.empty_like(x)
            todo = (margin_min > 0.0).nonzero().flatten()           # the one host sync per query
            for it in range(self.n_queries):
                m = int(todo.numel())
                p = p_selection(it, self.n_queries, self.p_init, self.rescale_schedule)
                s = min(max(int(round(math.sqrt(p * n_features / c))), 1), min(h, w))
                # the reference's draws, in its order, on the CPU generator; only the signs travel to the device
                vh = int((0 + (h - s - 0) * torch.rand([1])).long())
                vw = int((0 + (w - s - 0) * torch.rand([1])).long())
                sign = torch.sign(2 * torch.rand([c, 1, 1]) - 1).reshape(c).to(x.device, non_blocking=True)
                if m > 0:
                    x_new = x_new_buf[:m]
                    L.check(lib.rvlm_square_linf_propose(x.data_ptr(), x_best.data_ptr(), todo.data_ptr(), m, c, h, w,
                                                         vh, vw, s, eps, sign.data_ptr(), x_new.data_ptr(),
                                                         L.stream_ptr()), "rvlm_square_linf_propose")
                    margin, loss = self.margin_and_loss(x_new, y[todo])
                    better = (loss < loss_min[todo]).float()
                    loss_min[todo] = better * loss + (1. - better) * loss_min[todo]
                    take = torch.max(better, (margin <= 0.).float()).contiguous()
                    margin_min[todo] = take * margin + (1. - take) * margin_min[todo]
                    L.check(lib.rvlm_square_accept(x_best.data_ptr(), x_new.data_ptr(), todo.data_ptr(),
                                                   take.data_ptr(), m, n_features, L.stream_ptr()), "rvlm_square_accept")
                    n_queries[todo] += 1.
                todo = (margin_min > 0.0).nonzero().flatten()
                if self.verbose and todo.numel() != n:
                    done = (margin_min <= 0.).nonzero().flatten()
                    print('{}'.format(it + 1), '- success rate={}/{} ({:.2%})'.format(done.numel(), n, done.numel() / n),
                          '- avg # queries={:.1f}'.format(n_queries[done].mean().item()),
                          '- med # queries={:.1f}'.format(n_queries[done].median().item()),
                          '- loss={:.3f}'.format(loss_min.mean()))
                if todo.numel() == 0:
                    break
        return n_queries, x_best

    def perturb(self, x, y=None):
        """:param x: clean images  :param y: untargeted: clean labels (None -> the predicted ones); targeted: target
        labels (None -> random classes different from the prediction)   (square.py:549-618)"""
        _require_cuda(x, "x")
        self.init_hyperparam(x)
        adv = x.clone()
        with torch.no_grad():
            if y is None:
                out = self.predict(x)
                pred = out.max(1)[1]
                y = self.random_target_classes(pred, out.shape[-1]) if self.targeted \
                    else pred.detach().clone().long().to(self.device)
            else:
                y = y.detach().clone().long().to(self.device)
            hit = self.predict(x).max(1)[1] == y
        acc = ~hit if self.targeted else hit
        t_start = time.time()
        torch.random.manual_seed(self.seed)
        torch.cuda.random.manual_seed(self.seed)
        for r in range(self.n_restarts):
            todo = acc.nonzero().flatten()
            if todo.numel() == 0:
                continue
            _, cand = self.attack_single_run(x[todo].clone(), y[todo].clone())
            with torch.no_grad():
                hit = self.predict(cand).max(1)[1] == y[todo]
            still = ~hit if self.targeted else hit
            broken = (still == 0).nonzero().flatten()
            acc[todo[broken]] = 0
            adv[todo[broken]] = cand[broken].clone()
            if self.verbose:
                print('restart {} - robust accuracy: {:.2%}'.format(r, acc.float().mean()),
                      '- cum. time: {:.1f} s'.format(time.time() - t_start))
        return adv
