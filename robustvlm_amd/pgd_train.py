"""Drop-in ``pgd`` (train/pgd_train.py:5-68) on the MI355X-native path.

Same signature, same semantics, same exceptions.  Two execution routes, both on the GPU:

* fused  - ``forward`` is a :class:`ClipVisionModel` over the native engine and ``loss_fn`` a
  :class:`ComputeLossWrapper` (what train_one_epoch passes, …clip.py:300-321): the whole loop
  (encoder forward, loss, input-gradient backward, Linf update) is ONE call into librvlm.so
  (rvlm_pgd_run) with no host synchronisation; the reference's per-iteration asserts
  (:24, :40-42, :60-63) are accumulated in a device flag word and raised once after the loop.
* generic - any differentiable ``forward`` / ``loss_fn``: torch autograd supplies the gradient, the
  fused HIP kernel rvlm_pgd_linf_update does sign/momentum/step/project/clamp in one pass.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .attack_utils import LINF, L2, normalize_grad, project_perturbation
from .clip_model import ClipVisionModel, ComputeLossWrapper
from .engine import _require_cuda, _f32c


def _raise_from_flags(flags: int):
    if flags & L.FLAG_NAN_GRAD:
        print("attention: nan in gradient")           # pgd_train.py:41 (not an error; NaNs were zeroed)
    assert not (flags & L.FLAG_INPUT_RANGE), "data_clean is not in image space [0,1]"   # :24
    assert not (flags & L.FLAG_NAN_DELTA), "nan in perturbation"                         # :60
    assert not (flags & L.FLAG_ADV_RANGE), "data_clean + perturbation left [0,1]"        # :61-63


def pgd(forward, loss_fn, data_clean, targets, norm, eps, iterations, stepsize, output_normalize,
        perturbation=None, mode='min', momentum=0.9, verbose=False):
    """Minimize or maximize given loss (signature of train/pgd_train.py:5-19)."""
    _require_cuda(data_clean, "data_clean")
    if mode not in ("min", "max"):
        raise ValueError(f"Unknown mode: {mode}")                                        # :54
    if norm not in LINF and norm not in L2:
        raise NotImplementedError(f"Norm {norm} not supported")                          # utils.py:16
    lib = L.load()

    fused = (isinstance(forward, ClipVisionModel)
             and isinstance(loss_fn, ComputeLossWrapper) and not verbose
             and loss_fn.reduction in ("mean", "none") and data_clean.shape[0] > 1)
    if fused:
        kind, ref = loss_fn.fused_spec()
        x_adv, flags, _ = forward.model.pgd_run(
            data_clean, perturbation, kind, loss_fn.reduction, ref, targets, output_normalize, eps,
            iterations, stepsize, momentum, mode, loss_fn.logit_scale, norm_kind=0 if norm in LINF else 2)
        _raise_from_flags(int(flags.item()))          # the loop's only host sync
        return x_adv

    # ---- generic route: autograd for the model, fused HIP kernel for the update -----------------
    x = _f32c(data_clean)
    n = x.numel()
    flags = torch.zeros(1, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.rvlm_check_image_range(x.data_ptr(), n, flags.data_ptr(), L.stream_ptr()))
    delta = torch.zeros_like(x) if perturbation is None else _f32c(perturbation).clone()
    velocity = torch.zeros_like(x)
    for i in range(iterations):
        p = delta.detach().requires_grad_(True)
        with torch.enable_grad():
            out = forward(data_clean + p, output_normalize=output_normalize)
            loss = loss_fn(out, targets)
            if verbose:
                print(f'[{i}] {loss.item():.5f}')
        with torch.no_grad():
            gradient = _f32c(torch.autograd.grad(loss, p)[0])
            if norm in LINF:
                with torch.cuda.device(x.device):
                    L.check(lib.rvlm_pgd_linf_update(
                        x.data_ptr(), gradient.data_ptr(), delta.data_ptr(), velocity.data_ptr(), n,
                        float(eps), float(stepsize), float(momentum), 1 if mode == "max" else 0, None,
                        flags.data_ptr(), L.stream_ptr()), "rvlm_pgd_linf_update")
            else:   # L2 branch (utils.py:12-14,22-26): per-sample normalise / momentum / renorm / clamp in one kernel
                with torch.cuda.device(x.device):
                    L.check(lib.rvlm_pgd_l2_update(
                        x.data_ptr(), gradient.data_ptr(), delta.data_ptr(), velocity.data_ptr(), x[0].numel(),
                        x.shape[0], float(eps), float(stepsize), float(momentum), 1 if mode == "max" else 0, None,
                        flags.data_ptr(), L.stream_ptr()), "rvlm_pgd_l2_update")
    _raise_from_flags(int(flags.item()))
    return data_clean + delta.detach()
