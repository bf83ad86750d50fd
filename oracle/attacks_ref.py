"""CPU oracle: PGD / APGD perturbation loops (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, as explicit per-element fp32 arithmetic (numpy float32: one IEEE rounding per op, no FMA):

* ``pgd_ref``           <- train/pgd_train.py:5-68 (+ vlm_eval/attacks/utils.py:8-26), Linf and L2
* ``apgd_train_ref``    <- train/apgd_train.py:125-229, 284-355, 373 (Linf branch + controller)
* ``APGDAttackRef``     <- autoattack/autopgd_base.py:161-193, 205-451, 453-548 (Linf, CE loss)

Gradients come from torch CPU autograd through whatever ``forward`` / ``model`` callable is
passed (the oracle ViT in oracle/vit_ref.py, or any torch module).

Pinned by tests/golden/{pgd_*,apgd_*,autopgd_*}.npz = outputs of the reference's own functions
imported from /root/reference (tests/golden/make_golden.py); tests/test_oracle_golden.py demands
bit-equality.
"""
from __future__ import annotations

import math

import numpy as np
import torch

F32 = np.float32
LINF = ("inf", "linf", "Linf")
L2 = (2, 2.0, "l2", "L2", "2")


# --------------------------------------------------------------------------------------------------
# elementwise building blocks
# --------------------------------------------------------------------------------------------------
def sign_f32(a: np.ndarray) -> np.ndarray:
    """torch.sign semantics: sign(+-0) = 0 and sign(NaN) = 0 (numpy would give NaN)."""
    out = np.zeros_like(a, dtype=F32)
    out[a > 0] = 1.0
    out[a < 0] = -1.0
    return out


def normalize_grad_ref(g: np.ndarray, norm) -> np.ndarray:
    """vlm_eval/attacks/utils.py:19-26."""
    if norm in LINF:
        return sign_f32(g)
    if norm in L2:
        flat = g.reshape(g.shape[0], -1)
        # F.normalize: v / max(||v||_2, 1e-12); torch accumulates the norm in fp32
        nrm = torch.from_numpy(flat).norm(p=2, dim=1, keepdim=True).clamp_min(1e-12).numpy()
        return (flat / nrm).astype(F32).reshape(g.shape)
    raise NotImplementedError(f"Norm {norm} not supported")


def project_perturbation_ref(delta: np.ndarray, eps: float, norm) -> np.ndarray:
    """vlm_eval/attacks/utils.py:8-16 (eps is cast to fp32 at use)."""
    if norm in LINF:
        e = F32(eps)
        return np.minimum(np.maximum(delta, -e), e)
    if norm in L2:
        return torch.renorm(torch.from_numpy(delta), p=2, dim=0, maxnorm=eps).numpy()
    raise NotImplementedError(f"Norm {norm} not supported")


def pgd_linf_update_ref(x, g, delta, vel, eps, stepsize, momentum=0.9, mode="max"):
    """One PGD Linf update, train/pgd_train.py:38-59 (Appendix A.1).  Returns (delta, vel)."""
    g = np.where(np.isnan(g), F32(0), g).astype(F32)               # :40-42
    s = sign_f32(g)                                                # utils.py:21
    vel = sign_f32(F32(momentum) * vel + s)                        # :46-47
    stepv = F32(stepsize) * vel
    if mode == "max":
        delta = delta + stepv                                      # :52
    elif mode == "min":
        delta = delta - stepv                                      # :50
    else:
        raise ValueError(f"Unknown mode: {mode}")
    delta = project_perturbation_ref(delta, eps, "linf")           # :56
    delta = np.minimum(np.maximum(x + delta, F32(0)), F32(1)) - x  # :57-59
    return delta.astype(F32), vel.astype(F32)


def apgd_linf_step_ref(x, x_adv, x_adv_old, grad, step, a, eps):
    """One APGD Linf step (train/apgd_train.py:205-229 == autopgd_base.py:328-341), Appendix A.2.

    ``step`` is [B,1,1,1] fp32.  Returns (x_adv_new, x_adv_old_new)."""
    e = F32(eps)
    grad2 = x_adv - x_adv_old
    lo = x - e
    hi = x + e
    z = x_adv + step * sign_f32(grad)
    z = np.minimum(np.maximum(np.minimum(np.maximum(z, lo), hi), F32(0)), F32(1))
    u = x_adv + (z - x_adv) * F32(a) + grad2 * F32(1.0 - a)
    new = np.minimum(np.maximum(np.minimum(np.maximum(u, lo), hi), F32(0)), F32(1))
    return new.astype(F32), x_adv.copy()


def apgd_l2_step_ref(x, x_adv, x_adv_old, grad, step, a, eps, attack_form=False):
    """One APGD L2 step (train/apgd_train.py:231-254), with the reference's own tensor expressions (the per-sample
    norms are torch sums: ``(t ** 2).view(B, -1).sum(-1).sqrt()``, :16-20), so that the restatement is bit-equal to the
    reference on the CPU.  ``step`` is [B,1,1,1] fp32.  Returns (x_adv_new, x_adv_old_new).
    ``attack_form``: APGDAttack's spelling of the same step (autoattack/autopgd_base.py:343-351) - the gradient step is
    ``step * normalize(grad)`` = step * (g / (|g| + 1e-12)) where apgd_train writes (step * g) / (|g| + 1e-12): one fp32
    rounding apart."""
    xt, xa, xo, g = (torch.from_numpy(np.ascontiguousarray(t)) for t in (x, x_adv, x_adv_old, grad))
    st = torch.from_numpy(np.ascontiguousarray(step))

    def l2n(t):
        return (t ** 2).view(t.shape[0], -1).sum(-1).sqrt().view(-1, *[1] * (t.dim() - 1))

    grad2 = xa - xo
    x1 = xa + st * (g / (l2n(g) + 1e-12)) if attack_form else xa + st * g / (l2n(g) + 1e-12)
    x1 = torch.clamp(xt + (x1 - xt) / (l2n(x1 - xt) + 1e-12) * torch.min(eps * torch.ones_like(xt), l2n(x1 - xt)), 0.0, 1.0)
    x1 = xa + (x1 - xa) * a + grad2 * (1 - a)
    x1 = torch.clamp(xt + (x1 - xt) / (l2n(x1 - xt) + 1e-12) * torch.min(eps * torch.ones_like(xt), l2n(x1 - xt)), 0.0, 1.0)
    return x1.numpy().astype(F32), x_adv.copy()


def check_oscillation_ref(loss_steps: np.ndarray, j: int, k: int, k3: float = 0.75) -> np.ndarray:
    """train/apgd_train.py:117-122; negative row indices wrap like Python/torch indexing."""
    t = np.zeros(loss_steps.shape[1], dtype=F32)
    for c in range(k):
        t += (loss_steps[j - c] > loss_steps[j - c - 1]).astype(F32)
    return (t <= F32(k * k3) * np.ones_like(t)).astype(F32)


def apgd_schedule(n_iter: int):
    """train/apgd_train.py:153-156 / autopgd_base.py:157-159."""
    return (max(int(0.22 * n_iter), 1), max(int(0.06 * n_iter), 1), max(int(0.03 * n_iter), 1))


class ApgdController:
    """Per-sample best/oscillation/step-halving bookkeeping, train/apgd_train.py:320-355
    (same arithmetic as autopgd_base.py:402-446), Appendix A.3.  Host-side state machine over
    numpy arrays; mirrors what the device controller kernel does."""

    def __init__(self, n_iter, loss0, step0, rho=0.75):
        self.n_iter = n_iter
        self.rho = rho                                  # APGDAttack's thr_decr (autopgd_base.py:137,415-416)
        self.n_iter_2, self.n_iter_min, self.size_decr = apgd_schedule(n_iter)
        self.k = self.n_iter_2
        self.counter3 = 0
        B = loss0.shape[0]
        self.loss_steps = np.zeros((n_iter, B), dtype=F32)
        self.loss_best = loss0.astype(F32).copy()
        self.loss_best_last_check = self.loss_best.copy()
        self.reduced_last_check = np.ones(B, dtype=F32)
        self.step = step0.astype(F32).copy()            # [B]

    def update(self, i, loss_i, x_adv, grad, x_best, grad_best):
        """Mutates x_adv/grad/x_best/grad_best in place like the reference's index assignment."""
        y1 = loss_i.astype(F32)
        self.loss_steps[i] = y1
        ind = y1 > self.loss_best
        x_best[ind] = x_adv[ind]
        grad_best[ind] = grad[ind]
        self.loss_best[ind] = y1[ind]
        self.counter3 += 1
        red = None
        if self.counter3 == self.k:
            osc = check_oscillation_ref(self.loss_steps, i, self.k, self.rho)
            no_impr = (F32(1.0) - self.reduced_last_check) * \
                (self.loss_best_last_check >= self.loss_best).astype(F32)
            red = np.maximum(osc, no_impr)
            self.reduced_last_check = red.copy()
            self.loss_best_last_check = self.loss_best.copy()
            if red.sum() > 0:
                m = red > 0
                self.step[m] = self.step[m] / F32(2.0)
                x_adv[m] = x_best[m]
                grad[m] = grad_best[m]
            self.counter3 = 0
            self.k = max(self.k - self.size_decr, self.n_iter_min)
        return red


# --------------------------------------------------------------------------------------------------
# pgd  (train/pgd_train.py:5-68)
# --------------------------------------------------------------------------------------------------
def pgd_ref(forward, loss_fn, data_clean, targets, norm, eps, iterations, stepsize,
            output_normalize, perturbation=None, mode="min", momentum=0.9, verbose=False,
            trace=None):
    assert torch.max(data_clean) < 1. + 1e-6 and torch.min(data_clean) > -1e-6   # :24
    x = data_clean.detach().cpu().numpy().astype(F32)
    delta = np.zeros_like(x) if perturbation is None else \
        perturbation.detach().cpu().numpy().astype(F32)
    vel = np.zeros_like(x)
    for i in range(iterations):
        p = torch.from_numpy(delta.copy()).requires_grad_(True)
        with torch.enable_grad():
            out = forward(data_clean + p, output_normalize=output_normalize)      # :32
            loss = loss_fn(out, targets)                                          # :33
        g = torch.autograd.grad(loss, p)[0].numpy().astype(F32)                   # :38
        if trace is not None:
            trace.append(dict(loss=float(loss.detach()), grad=g.copy()))
        if norm in LINF:
            delta, vel = pgd_linf_update_ref(x, g, delta, vel, eps, stepsize, momentum, mode)
        else:
            g = np.where(np.isnan(g), F32(0), g).astype(F32)
            g = normalize_grad_ref(g, norm)
            vel = normalize_grad_ref((F32(momentum) * vel + g).astype(F32), norm)
            sgn = F32(1) if mode == "max" else F32(-1)
            if mode not in ("max", "min"):
                raise ValueError(f"Unknown mode: {mode}")
            delta = delta + sgn * (F32(stepsize) * vel)
            delta = project_perturbation_ref(delta.astype(F32), eps, norm)
            delta = (np.minimum(np.maximum(x + delta, F32(0)), F32(1)) - x).astype(F32)
        assert not np.isnan(delta).any()                                          # :60
        xa = x + delta
        assert xa.max() < 1. + 1e-6 and xa.min() > -1e-6                          # :61-63
    return torch.from_numpy((x + delta).astype(F32))                              # :68


# --------------------------------------------------------------------------------------------------
# apgd_train  (train/apgd_train.py:125-373, Linf)
# --------------------------------------------------------------------------------------------------
def _fwd_bwd(model_call, loss_call, x_adv_np, y, need_grad=True):
    xa = torch.from_numpy(x_adv_np.copy()).requires_grad_(need_grad)
    with torch.enable_grad():
        logits = model_call(xa)
        loss_indiv = loss_call(logits, y)
        loss = loss_indiv.sum()
    grad = torch.autograd.grad(loss, [xa])[0].detach().numpy().astype(F32) if need_grad else None
    return logits.detach(), loss_indiv.detach().numpy().astype(F32), grad


def apgd_train_ref(model, x, y, norm, eps, n_iter=10, use_rs=False, loss_fn=None, verbose=False,
                   is_train=True, initial_stepsize=None, trace=None):
    assert not model.training                                                     # :127
    norm = norm.replace("linf", "Linf").replace("l2", "L2")
    if norm not in ("Linf", "L2"):
        raise NotImplementedError("oracle restates the Linf and L2 branches (SURVEY.md 8(a4))")
    step_fn = apgd_linf_step_ref if norm == "Linf" else apgd_l2_step_ref
    if use_rs:
        raise NotImplementedError  # reference raises too (:132-135)
    xn = x.detach().cpu().numpy().astype(F32)
    B = xn.shape[0]
    x_adv = np.minimum(np.maximum(xn, F32(0)), F32(1))                            # :137
    x_best = x_adv.copy()
    x_best_adv = x_adv.copy()
    alpha = 2.0
    if initial_stepsize:
        alpha = initial_stepsize / eps                                            # :168-169
    step0 = np.full((B,), F32(alpha * eps), dtype=F32)                            # :171-174
    call = lambda t: model(t, output_normalize=True)                              # noqa: E731 (:181)
    logits, loss_indiv, grad = _fwd_bwd(call, loss_fn, x_adv, y)                  # :177-190
    grad_best = grad.copy()
    acc = (logits.max(1)[1] == y).numpy()                                         # :192
    ctl = ApgdController(n_iter, loss_indiv, step0)
    x_adv_old = x_adv.copy()
    for i in range(n_iter):
        a = 0.75 if i > 0 else 1.0
        x_adv, x_adv_old = step_fn(xn, x_adv, x_adv_old, grad, ctl.step.reshape(B, 1, 1, 1), a, eps)
        last = (i == n_iter - 1)
        logits, loss_indiv, g_new = _fwd_bwd(call, loss_fn, x_adv, y, need_grad=not last)
        if not last:
            grad = g_new                                                          # :293-295
        pred = (logits.max(1)[1] == y).numpy()
        acc = np.minimum(acc, pred)
        x_best_adv[~pred] = x_adv[~pred]                                          # :304-305
        if trace is not None:
            trace.append(dict(loss=loss_indiv.copy(), step=ctl.step.copy(), x_adv=x_adv.copy()))
        ctl.update(i, loss_indiv, x_adv, grad, x_best, grad_best)                 # :320-355
    return torch.from_numpy(x_best_adv)                                           # :373


# --------------------------------------------------------------------------------------------------
# APGDAttack (autoattack/autopgd_base.py, Linf / L2; CE / DLR / targeted DLR)
# --------------------------------------------------------------------------------------------------
class APGDAttackRef:
    def __init__(self, predict, n_iter=100, norm="Linf", n_restarts=1, eps=None, seed=0, loss="ce",
                 eot_iter=1, rho=.75, topk=None, verbose=False, device=None, use_largereps=False,
                 is_tf_model=False, logger=None, alpha=None, use_rs=True):
        assert norm in ("Linf", "L2") and loss in ("ce", "dlr", "dlr-targeted") and eot_iter == 1 and not use_largereps \
            and not is_tf_model, "oracle restates the Linf / L2 paths with the CE / DLR / targeted-DLR losses (SURVEY.md 8(a12), 8(f3))"
        assert eps is not None
        self.norm = norm
        self.model, self.n_iter, self.eps, self.n_restarts = predict, n_iter, eps, n_restarts
        self.seed, self.thr_decr, self.alpha, self.use_rs = seed, rho, alpha, use_rs
        self.loss, self.y_target = loss, None

    def dlr_loss(self, x, y):
        # autopgd_base.py:195-201
        x_sorted, ind_sorted = x.sort(dim=1)
        ind = (ind_sorted[:, -1] == y).float()
        u = torch.arange(x.shape[0])
        return -(x[u, y] - x_sorted[:, -2] * ind - x_sorted[:, -1] * (1. - ind)) / (
            x_sorted[:, -1] - x_sorted[:, -3] + 1e-12)

    def dlr_loss_targeted(self, x, y):
        # autopgd_base.py:613-618
        x_sorted, _ = x.sort(dim=1)
        u = torch.arange(x.shape[0])
        return -(x[u, y] - x[u, self.y_target]) / (x_sorted[:, -1] - .5 * (x_sorted[:, -3] + x_sorted[:, -4]) + 1e-12)

    def _criterion(self):
        # autopgd_base.py:243-254
        if self.loss == "ce":
            return lambda lg, yy: torch.nn.functional.cross_entropy(lg, yy, reduction="none")
        return self.dlr_loss if self.loss == "dlr" else self.dlr_loss_targeted

    def _random_start(self, xn):
        # autopgd_base.py:210-214 + normalize() :180-183: x + eps * t / (max|t| + 1e-12), t~U(-1,1)
        # The reference draws t on the CPU generator and moves it to the device.
        if self.norm == "L2":
            # :184-185, 215-218: t ~ N(0,1), x + eps * t / (|t|_2 + 1e-12), in the reference's own torch expressions
            t = torch.randn(xn.shape)
            xt = torch.from_numpy(xn)
            nrm = (t ** 2).view(t.shape[0], -1).sum(-1).sqrt()
            return (xt + self.eps * torch.ones_like(xt) * (t / (nrm.view(-1, 1, 1, 1) + 1e-12))).numpy().astype(F32)
        t = (2 * torch.rand(xn.shape) - 1).numpy().astype(F32)
        tmax = np.abs(t).reshape(t.shape[0], -1).max(1).reshape(-1, 1, 1, 1)
        return (xn + F32(self.eps) * np.ones_like(xn) * (t / (tmax + F32(1e-12)))).astype(F32)

    def attack_single_run(self, x, y, x_init=None):
        xn = x.detach().cpu().numpy().astype(F32)
        B = xn.shape[0]
        x_adv = self._random_start(xn) if self.use_rs else xn.copy()
        if x_init is not None:
            x_adv = x_init.detach().cpu().numpy().astype(F32).copy()
        x_adv = np.minimum(np.maximum(x_adv, F32(0)), F32(1))                     # :233
        x_best, x_best_adv = x_adv.copy(), x_adv.copy()
        ce = self._criterion()
        logits, loss_indiv, grad = _fwd_bwd(self.model, ce, x_adv, y)             # :267-286
        grad_best = grad.copy()
        acc = (logits.max(1)[1] == y).numpy()
        alpha = 2.0 if self.alpha is None else self.alpha                         # :296-299
        ctl = ApgdController(self.n_iter, loss_indiv, np.full((B,), F32(alpha * self.eps), F32), rho=self.thr_decr)
        x_adv_old = x_adv.copy()
        for i in range(self.n_iter):
            a = 0.75 if i > 0 else 1.0
            if self.norm == "L2":
                x_adv, x_adv_old = apgd_l2_step_ref(xn, x_adv, x_adv_old, grad, ctl.step.reshape(B, 1, 1, 1), a, self.eps,
                                                    attack_form=True)
            else:
                x_adv, x_adv_old = apgd_linf_step_ref(xn, x_adv, x_adv_old, grad,
                                                      ctl.step.reshape(B, 1, 1, 1), a, self.eps)
            logits, loss_indiv, grad = _fwd_bwd(self.model, ce, x_adv, y)         # :369-386
            pred = (logits.max(1)[1] == y).numpy()
            acc = np.minimum(acc, pred)
            x_best_adv[~pred] = x_adv[~pred]                                      # :388-392
            ctl.update(i, loss_indiv, x_adv, grad, x_best, grad_best)             # :402-446
        return (torch.from_numpy(x_best), torch.from_numpy(acc), torch.from_numpy(ctl.loss_best),
                torch.from_numpy(x_best_adv))

    def perturb(self, x, y=None, best_loss=False, x_init=None):
        assert not best_loss, "oracle restates best_loss=False only"
        x = x.detach().clone().float()
        y_pred = self.model(x).max(1)[1]                                          # :469
        y = y_pred.detach().clone().long() if y is None else y.detach().clone().long()
        adv = x.clone()
        acc = y_pred == y
        torch.random.manual_seed(self.seed)                                       # :505
        for _ in range(self.n_restarts):                                          # :508
            ind = acc.nonzero().squeeze(1)
            if ind.numel() == 0:
                continue
            _, acc_curr, _, adv_curr = self.attack_single_run(x[ind].clone(), y[ind].clone())
            fooled = (acc_curr == 0).nonzero().squeeze(1)
            acc[ind[fooled]] = False
            adv[ind[fooled]] = adv_curr[fooled].clone()
        return adv
