"""Drop-in ``APGDAttack`` / ``APGDAttack_targeted`` (autoattack/autopgd_base.py:89-582, 584-707) for BASELINE
config 5: L-inf (and L2, the other norm ``CLIP_eval/clip_robustbench.py --norm`` offers), CE / DLR / targeted-DLR losses,
random start, restarts over still-correct points.

Native routes as in apgd_train.py: fused (predict is a :class:`ClassificationModel` over the
engine -> rvlm_apgd_run with the zero-shot head and the loss on the device) or generic (any ``predict``).
Branches the repo's configs never select (L1 / TF adapters / use_largereps / EOT) raise NotImplementedError
(SURVEY.md section 2, row 7).  L2: the per-sample norms are deterministic fp32 sums in the kernels' own order, and the
gradient step is evaluated as ``(step * g) / |g|`` (the form of train/apgd_train.py:232) where autopgd_base.py:344 writes
``step * (g / |g|)`` - one rounding apart, below the spread of the sums themselves.
"""
from __future__ import annotations

import time

import torch

from . import _lib as L
from .apgd_train import _apgd_linf_generic, apgd_schedule
from .clip_model import ClassificationModel, _CeLogitsFn
from .engine import _require_cuda, _f32c


class APGDAttack():
    """AutoPGD https://arxiv.org/abs/2003.01690 (constructor of autopgd_base.py:105-124)."""

    def __init__(self, predict, n_iter=100, norm='Linf', n_restarts=1, eps=None, seed=0, loss='ce',
                 eot_iter=1, rho=.75, topk=None, verbose=False, device=None, use_largereps=False,
                 is_tf_model=False, logger=None, alpha=None, use_rs=True):
        self.model = predict
        self.n_iter = n_iter
        self.eps = eps
        self.norm = norm
        self.n_restarts = n_restarts
        self.seed = seed
        self.loss = loss
        self.eot_iter = eot_iter
        self.thr_decr = rho
        self.topk = topk
        self.verbose = verbose
        self.device = device
        self.use_rs = use_rs
        self.use_largereps = use_largereps
        self.n_iter_orig = n_iter + 0
        assert self.norm in ['Linf', 'L2', 'L1']
        assert not self.eps is None
        self.eps_orig = eps + 0.
        self.is_tf_model = is_tf_model
        self.y_target = None
        self.logger = logger
        self.alpha = alpha
        self.n_iter_2, self.n_iter_min, self.size_decr = apgd_schedule(self.n_iter)
        if norm not in ('Linf', 'L2') or is_tf_model or use_largereps or eot_iter != 1:
            raise NotImplementedError("native APGDAttack covers norm='Linf' / 'L2', eot_iter=1, torch models")
        self._norm_kind = 0 if norm == 'Linf' else 2

    def init_hyperparam(self, x):
        if self.device is None:
            self.device = x.device
        self.orig_dim = list(x.shape[1:])
        self.ndims = len(self.orig_dim)
        if self.seed is None:
            self.seed = time.time()

    def _random_start(self, x):
        """Linf: x + eps * t / max|t|, t ~ U(-1,1);  L2: x + eps * t / |t|_2, t ~ N(0,1).  t is drawn on the CPU generator
        like the reference (autopgd_base.py:210-218 does torch.rand / torch.randn(x.shape).to(device)), normalised on the
        device."""
        lib = L.load()
        if self.norm == 'Linf':
            t = (2 * torch.rand(x.shape).to(x.device).detach() - 1).contiguous()
            fn = lib.rvlm_linf_random_start
        else:
            t = torch.randn(x.shape).to(x.device).detach().contiguous()
            fn = lib.rvlm_l2_random_start
        out = torch.empty_like(x)
        B = x.shape[0]
        with torch.cuda.device(x.device):
            L.check(fn(x.data_ptr(), t.data_ptr(), float(self.eps), x[0].numel(), B, out.data_ptr(), L.stream_ptr()))
        return out

    # ---- the DLR losses for the generic (arbitrary ``predict``) route; same values as autopgd_base.py:195-201 and
    # :613-618, written with top-k instead of a full sort
    def dlr_loss(self, x, y):
        top, idx = x.topk(3, dim=1)
        zy = x.gather(1, y.view(-1, 1)).squeeze(1)
        ind = (idx[:, 0] == y).float()
        return -(zy - top[:, 1] * ind - top[:, 0] * (1. - ind)) / (top[:, 0] - top[:, 2] + 1e-12)

    def dlr_loss_targeted(self, x, y):
        top, _ = x.topk(4, dim=1)
        zy = x.gather(1, y.view(-1, 1)).squeeze(1)
        zt = x.gather(1, self.y_target.view(-1, 1)).squeeze(1)
        return -(zy - zt) / (top[:, 0] - .5 * (top[:, 2] + top[:, 3]) + 1e-12)

    def attack_single_run(self, x, y, x_init=None):
        """Returns (x_best, acc, loss_best, x_best_adv) like autopgd_base.py:205-451."""
        if self.loss not in ('ce', 'dlr', 'dlr-targeted'):
            raise ValueError('unknowkn loss')                                 # autopgd_base.py:253-254
        x = _f32c(x)
        start = self._random_start(x) if self.use_rs else None
        if x_init is not None:
            start = _f32c(x_init)
        alpha = 2. if self.alpha is None else self.alpha                    # :296-299
        step0 = alpha * self.eps
        m = self.model
        if isinstance(m, ClassificationModel) and x.shape[0] > 1 and m.logit_scale and m._identity_resizer:
            # every quantity of the attack is per sample: batches beyond the engine's workspace (AutoAttack's bs = 250 on
            # a max_batch = 128 engine) run as chunks of the device loop (chunks of >= 2 images: the losses need batch > 1)
            B, mb = x.shape[0], m.model.max_batch
            bounds = list(range(0, B, mb)) + [B]
            if len(bounds) > 2 and bounds[-1] - bounds[-2] < 2:
                bounds[-2] -= 1
            yt = self.y_target if self.loss == 'dlr-targeted' else None
            parts = []
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                parts.append(m.model.apgd_run(
                    x[lo:hi], None if start is None else start[lo:hi], self.loss, m.text_embedding, y[lo:hi], True,
                    self.eps, self.n_iter, step0, train_variant=False, logits_from_head=True,
                    logit_scale=m.logit_scale_value, want_extra=True, y_target=None if yt is None else yt[lo:hi],
                    rho=self.thr_decr, norm_kind=self._norm_kind))
            x_best_adv, x_best, loss_best, acc = (torch.cat([p[i] for p in parts]) for i in range(4))
            return x_best, acc.bool(), loss_best, x_best_adv
        if self.loss == 'ce':
            # nn.CrossEntropyLoss(reduction='none') (autopgd_base.py:249) on rvlm_ce_logits: unlike the trainer's ce() it
            # has no batch > 1 assert - a restart can be left with ONE robust point (autopgd_base.py:494-500)
            crit = lambda lg, yy: _CeLogitsFn.apply(lg, yy, L.RED_NONE)   # noqa: E731
        else:
            crit = self.dlr_loss if self.loss == 'dlr' else self.dlr_loss_targeted
        return _apgd_linf_generic(self.model, crit, x, y, self.eps, self.n_iter, step0, False, x_init=start,
                                  rho=self.thr_decr, norm_kind=self._norm_kind)

    # ---- shared pieces of the two perturb() flavours --------------------------------------------------------------
    def _setup(self, x, y):
        """Inputs on the attack device, labels (the clean prediction when y is None) and the mask of points that are
        still classified correctly (autopgd_base.py:462-475 and :637-650 do the same preparation)."""
        _require_cuda(x, "x")
        if y is not None and y.dim() == 0:
            x.unsqueeze_(0)
            y.unsqueeze_(0)
        self.init_hyperparam(x)
        x = x.detach().clone().float().to(self.device)
        with torch.no_grad():
            pred = self.model(x).max(1)[1]
        y = (pred if y is None else y).detach().clone().long().to(self.device)
        return x, y, pred == y

    def _banner(self, acc):
        if self.verbose:
            print('-------------------------- ', 'running {}-attack with epsilon {:.5f}'.format(self.norm, self.eps),
                  '--------------------------')
            print('initial accuracy: {:.2%}'.format(acc.float().mean()))

    def _reseed(self):
        torch.random.manual_seed(self.seed)
        torch.cuda.random.manual_seed(self.seed)

    def _attack_survivors(self, x, y, acc, adv, pick_target=None):
        """One run over the points that are still robust; fooled points leave ``acc`` and get their adversarial
        example written into ``adv`` (the loop body of autopgd_base.py:490-507 / :660-686)."""
        todo = acc.nonzero().squeeze(1)
        if todo.numel() == 0:
            return
        xs, ys = x[todo].clone(), y[todo].clone()
        if pick_target is not None:
            self.y_target = pick_target(xs)
        _, still, _, cand = self.attack_single_run(xs, ys)
        broken = (still == 0).nonzero().squeeze(1)
        acc[todo[broken]] = 0
        adv[todo[broken]] = cand[broken].clone()

    def perturb(self, x, y=None, best_loss=False, x_init=None):
        """:param best_loss: if True the points attaining highest loss are returned, otherwise
        adversarial examples (autopgd_base.py:453-548)."""
        assert self.loss in ['ce', 'dlr']
        x, y, acc = self._setup(x, y)
        self._banner(acc)
        t_start = time.time()
        if best_loss:
            best = x.detach().clone()
            best_val = torch.full([x.shape[0]], -float('inf'), device=self.device)
            for r in range(self.n_restarts):
                cand, _, val, _ = self.attack_single_run(x, y)
                better = (val > best_val).nonzero().squeeze(1)
                best[better] = cand[better] + 0.
                best_val[better] = val[better] + 0.
                if self.verbose:
                    print('restart {} - loss: {:.5f}'.format(r, best_val.sum()))
            return best
        adv = x.clone()
        self._reseed()
        for r in range(self.n_restarts):
            self._attack_survivors(x, y, acc, adv)
            if self.verbose:
                print('restart {} - robust accuracy: {:.2%}'.format(r, acc.float().mean()),
                      '- cum. time: {:.1f} s'.format(time.time() - t_start))
        return adv


class APGDAttack_targeted(APGDAttack):
    """AutoPGD on the targeted DLR loss (autopgd_base.py:584-707): for each of the ``n_target_classes`` most likely
    wrong classes, one run over the points that are still classified correctly."""

    def __init__(self, predict, n_iter=100, norm='Linf', n_restarts=1, eps=None, seed=0, eot_iter=1, rho=.75,
                 topk=None, n_target_classes=9, verbose=False, device=None, use_largereps=False, is_tf_model=False,
                 logger=None, alpha=None, use_rs=True):
        super().__init__(predict, n_iter=n_iter, norm=norm, n_restarts=n_restarts, eps=eps, seed=seed,
                         loss='dlr-targeted', eot_iter=eot_iter, rho=rho, topk=topk, verbose=verbose, device=device,
                         use_largereps=use_largereps, is_tf_model=is_tf_model, logger=logger, alpha=alpha, use_rs=use_rs)
        self.y_target = None
        self.n_target_classes = n_target_classes

    def perturb(self, x, y=None, x_init=None):
        """:param x: clean images  :param y: clean labels, if None we use the predicted labels"""
        assert self.loss in ['dlr-targeted']
        x, y, acc = self._setup(x, y)
        adv = x.clone()
        self._banner(acc)
        t_start = time.time()
        self._reseed()
        for rank in range(2, self.n_target_classes + 2):        # the rank-th most likely class of each point

            def kth_class(xs, rank=rank):
                with torch.no_grad():
                    return self.model(xs).sort(dim=1)[1][:, -rank]

            for r in range(self.n_restarts):
                self._attack_survivors(x, y, acc, adv, pick_target=kth_class)
                if self.verbose:
                    print('target class {}'.format(rank),
                          '- restart {} - robust accuracy: {:.2%}'.format(r, acc.float().mean()),
                          '- cum. time: {:.1f} s'.format(time.time() - t_start))
        return adv
