"""Drop-in ``apgd_train`` (train/apgd_train.py:125-373) on the MI355X-native path (Linf).

Fused route (ClipVisionModel + ComputeLossWrapper): one rvlm_apgd_run call - step, encoder
forward/backward, per-sample controller and restart-from-best all stay on the device.  Generic
route: autograd for the model, HIP kernels rvlm_apgd_linf_step / _controller / _select for the
per-sample arithmetic (no nonzero()/index host syncs, apgd_train.py:304,323,347).
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from .clip_model import ClipVisionModel, ComputeLossWrapper
from .engine import _require_cuda, _f32c


def apgd_schedule(n_iter: int):
    """(n_iter_2, n_iter_min, size_decr), apgd_train.py:153-156."""
    return max(int(0.22 * n_iter), 1), max(int(0.06 * n_iter), 1), max(int(0.03 * n_iter), 1)


def _apgd_linf_generic(model_call, loss_call, x, y, eps, n_iter, step0, train_variant, x_init=None, norm_kind=0,
                       rho=0.75):
    """Shared host loop of apgd_train / APGDAttack.attack_single_run for arbitrary callables.  ``rho``: the oscillation
    threshold (APGDAttack's parameter, autopgd_base.py:111,137,415-416; apgd_train.py:117,334 hard-codes 0.75).
    Returns (x_best, acc(bool), loss_best, x_best_adv)."""
    lib = L.load()
    x = _f32c(x)
    B = x.shape[0]
    npix = x[0].numel()
    dev = x.device
    st = L.stream_ptr

    def u8(n):
        return torch.zeros(n, dtype=torch.uint8, device=dev)

    x_adv = (x if x_init is None else _f32c(x_init)).clamp(0., 1.).contiguous().clone()
    x_best, x_best_adv, x_adv_old = x_adv.clone(), x_adv.clone(), x_adv.clone()
    loss_steps = torch.zeros(n_iter, B, dtype=torch.float32, device=dev)
    pred = u8(B)

    def evaluate(need_grad):
        xa = x_adv.detach().clone().requires_grad_(need_grad)
        with torch.enable_grad():
            logits = model_call(xa)
            loss_indiv = loss_call(logits, y)
            loss = loss_indiv.sum()
        g = _f32c(torch.autograd.grad(loss, [xa])[0]) if need_grad else None
        lg = _f32c(logits)
        with torch.cuda.device(dev):
            L.check(lib.rvlm_argmax_eq(lg.data_ptr(), y.data_ptr(), B, lg.shape[1], pred.data_ptr(), st()))
        return _f32c(loss_indiv), g

    y = y.detach().to(torch.int64).contiguous()
    loss_indiv, grad = evaluate(True)
    grad_best = grad.clone()
    acc = pred.clone()
    loss_best = loss_indiv.clone()
    loss_best_lc = loss_best.clone()
    reduced_lc = torch.ones_like(loss_best)
    step = torch.full((B,), float(step0), dtype=torch.float32, device=dev)
    f0, f1, f2 = u8(B), u8(B), u8(B)
    k, n_iter_min, size_decr = apgd_schedule(n_iter)
    counter3 = 0
    for i in range(n_iter):
        a = 0.75 if i > 0 else 1.0
        with torch.cuda.device(dev):
            step_fn = lib.rvlm_apgd_l2_step if norm_kind == 2 else lib.rvlm_apgd_linf_step
            L.check(step_fn(x.data_ptr(), x_adv.data_ptr(), x_adv_old.data_ptr(),
                            grad.data_ptr(), step.data_ptr(), a, float(eps), npix, B, st()))
        need_grad = not (train_variant and i == n_iter - 1)
        loss_indiv, g = evaluate(need_grad)
        if need_grad:
            grad = g
        counter3 += 1
        do_check = int(counter3 == k)
        with torch.cuda.device(dev):
            L.check(lib.rvlm_apgd_controller_rho(i, B, n_iter, k, do_check, float(rho), loss_indiv.data_ptr(), pred.data_ptr(),
                                             loss_steps.data_ptr(), loss_best.data_ptr(), loss_best_lc.data_ptr(),
                                             reduced_lc.data_ptr(), step.data_ptr(), acc.data_ptr(),
                                             f0.data_ptr(), f1.data_ptr(), f2.data_ptr(), st()))
            L.check(lib.rvlm_apgd_select(x_adv.data_ptr(), grad.data_ptr(), x_best.data_ptr(),
                                         grad_best.data_ptr(), x_best_adv.data_ptr(), f0.data_ptr(),
                                         f1.data_ptr(), f2.data_ptr(), npix, B, st()))
        if do_check:
            counter3 = 0
            k = max(k - size_decr, n_iter_min)
    return x_best, acc.bool(), loss_best, x_best_adv


def apgd_train(model, x, y, norm, eps, n_iter=10, use_rs=False, loss_fn=None, verbose=False,
               is_train=True, initial_stepsize=None):
    """Signature of train/apgd_train.py:125-126; returns x_best_adv (:373)."""
    assert not model.training                                              # :127
    _require_cuda(x, "x")
    norm = norm.replace('linf', 'Linf').replace('l2', 'L2')
    if use_rs:
        raise NotImplementedError                                          # reference raises (:132-135)
    if norm not in ('Linf', 'L2'):
        raise NotImplementedError(f"apgd_train on the native path covers norm='Linf' and 'L2' (got {norm}); "
                                  f"SURVEY.md 8(a4)")
    norm_kind = 0 if norm == 'Linf' else 2
    alpha = 2.
    if initial_stepsize:
        alpha = initial_stepsize / eps                                     # :168-169
    step0 = alpha * eps                                                    # double, cast to fp32 at use
    if isinstance(model, ClipVisionModel) and isinstance(loss_fn, ComputeLossWrapper) \
            and isinstance(y, torch.Tensor) and not verbose and x.shape[0] > 1:
        kind, ref = loss_fn.fused_spec()
        # apgd always applies output normalization (:181,288); the argmax test runs on the model
        # output itself, i.e. the embedding
        x_best_adv, _, _, _ = model.model.apgd_run(x, None, kind, ref, y, True, eps, n_iter, step0,
                                                   train_variant=True, logits_from_head=False,
                                                   logit_scale=loss_fn.logit_scale, norm_kind=norm_kind)
        return x_best_adv
    call = lambda t: model(t, output_normalize=True)   # noqa: E731
    return _apgd_linf_generic(call, loss_fn, x, y, eps, n_iter, step0, True, norm_kind=norm_kind)[3]
