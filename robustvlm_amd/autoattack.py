"""AutoAttack orchestration for the robust-accuracy evaluation of BASELINE config 5.

Drop-in for ``autoattack.AutoAttack`` (autoattack/autoattack.py:12-330) as CLIP_eval/clip_robustbench.py:289-300 uses
it: ``AutoAttack(model, norm='Linf', eps=eps, version='custom', attacks_to_run=['apgd-ce', 'apgd-t'], alpha=...,
iterations_apgd=..., use_rs=...)`` then ``run_standard_evaluation(x, y, bs=...)``.  The attacks themselves are the
native :class:`~robustvlm_amd.autopgd.APGDAttack` / :class:`~robustvlm_amd.autopgd.APGDAttack_targeted` (whole loops on
the device when ``model`` is a :class:`~robustvlm_amd.clip_model.ClassificationModel` over the engine).

Built: ``apgd-ce``, ``apgd-dlr``, ``apgd-t`` and ``square`` (the black-box route of clip_robustbench.py:150-151, L-inf;
:class:`~robustvlm_amd.square.SquareAttack`).  NOT built (the reference's configs for this path never select them):
``fab``, ``fab-t`` -> NotImplementedError when they would run; their hyper-parameter holder exists so that
``set_version`` and user code that tweaks ``adversary.fab.n_restarts`` keep working.
"""
from __future__ import annotations

import json
import math
import sys
import time
import warnings
from datetime import datetime
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from .autopgd import APGDAttack, APGDAttack_targeted
from .square import SquareAttack

_BUILT = ("apgd-ce", "apgd-dlr", "apgd-t", "square")
_KNOWN = _BUILT + ("fab", "fab-t")


class EvaluationState:
    """Resumable bookkeeping of an evaluation (same JSON keys as autoattack/state.py, so state files interoperate)."""
    _SAVE_TIMEOUT = 60

    def __init__(self, attacks_to_run, path=None, run_attacks=(), robust_flags=None, clean_accuracy=float("nan")):
        self._attacks_to_run = set(attacks_to_run)
        self.path = Path(path) if path is not None else None
        self._run_attacks = set(run_attacks)
        self._robust_flags = robust_flags
        self._clean_accuracy = clean_accuracy
        self._last_saved = datetime(1, 1, 1)

    # -- persistence
    def to_disk(self, force=False):
        if self.path is None or (not force and (datetime.now() - self._last_saved).total_seconds() < self._SAVE_TIMEOUT):
            return
        self._last_saved = datetime.now()
        blob = {"_attacks_to_run": sorted(self._attacks_to_run), "path": str(self.path),
                "_run_attacks": sorted(self._run_attacks),
                "_robust_flags": None if self._robust_flags is None else self._robust_flags.cpu().tolist(),
                "_last_saved": self._last_saved.isoformat(), "_SAVE_TIMEOUT": self._SAVE_TIMEOUT,
                "_clean_accuracy": self._clean_accuracy}
        with self.path.open("w") as f:
            json.dump(blob, f, default=str)

    @classmethod
    def from_disk(cls, path):
        path = Path(path)
        with path.open("r") as f:
            blob = json.load(f)
        if Path(blob["path"]) != path:
            warnings.warn(UserWarning("The given path is different from the one found in the state file."))
        flags = blob["_robust_flags"]
        st = cls(blob["_attacks_to_run"], path=blob["path"], run_attacks=blob["_run_attacks"],
                 robust_flags=None if flags is None else torch.tensor(flags, dtype=torch.bool),
                 clean_accuracy=blob["_clean_accuracy"])
        st._last_saved = datetime.fromisoformat(blob["_last_saved"])
        return st

    # -- accessors (setting flags / accuracy persists immediately, like the reference's property setters)
    attacks_to_run = property(lambda self: self._attacks_to_run)
    run_attacks = property(lambda self: self._run_attacks)

    @property
    def robust_flags(self):
        return self._robust_flags

    @robust_flags.setter
    def robust_flags(self, flags):
        self._robust_flags = flags
        self.to_disk(force=True)

    @property
    def clean_accuracy(self):
        return self._clean_accuracy

    @clean_accuracy.setter
    def clean_accuracy(self, acc):
        self._clean_accuracy = acc
        self.to_disk(force=True)

    def add_run_attack(self, attack):
        self._run_attacks.add(attack)
        self.to_disk()

    @property
    def robust_accuracy(self):
        if self._robust_flags is None:
            raise ValueError("robust_flags is not set yet. Start the attack first.")
        if self._attacks_to_run - self._run_attacks:
            warnings.warn("You are checking `robust_accuracy` before all the attacks have been run.")
        return self._robust_flags.float().mean().item()


class _Log:
    def __init__(self, log_path=None):
        self.log_path = log_path

    def log(self, msg):
        print(msg)
        if self.log_path is not None:
            with open(self.log_path, "a") as f:
                f.write(msg + "\n")
                f.flush()


class AutoAttack():
    def __init__(self, model, norm='Linf', eps=.3, seed=None, verbose=True, attacks_to_run=[], version='standard',
                 is_tf_model=False, device='cuda', log_path=None, alpha=None, iterations_apgd=100, use_rs=True):
        assert norm in ['Linf', 'L2', 'L1']
        if is_tf_model:
            raise NotImplementedError("native AutoAttack drives torch callables only")
        if version in ['standard', 'plus', 'rand'] and attacks_to_run != []:
            raise ValueError("attacks_to_run will be overridden unless you use version='custom'")
        self.model, self.norm, self.epsilon, self.seed, self.verbose = model, norm, eps, seed, verbose
        self.attacks_to_run, self.version, self.is_tf_model, self.device = list(attacks_to_run), version, False, device
        self.logger = _Log(log_path)
        common = dict(n_iter=iterations_apgd, verbose=False, eps=eps, norm=norm, eot_iter=1, rho=.75, seed=seed,
                      device=device, logger=self.logger, alpha=alpha, use_rs=use_rs)
        self.apgd = APGDAttack(model, n_restarts=5, **common)                       # autoattack.py:34-36
        self.apgd_targeted = APGDAttack_targeted(model, n_restarts=1, **common)     # :47-49
        # hyper-parameter holders of the attacks that are not built (see module docstring)
        self.fab = SimpleNamespace(n_restarts=5, n_iter=100, n_target_classes=9, targeted=False, seed=seed)
        if norm == 'Linf':                                                          # autoattack.py:42-44
            self.square = SquareAttack(model, p_init=.8, n_queries=5000, eps=eps, norm=norm, n_restarts=1, seed=seed,
                                       verbose=False, device=device, resc_schedule=False)
        else:
            self.square = SimpleNamespace(p_init=.8, n_queries=5000, n_restarts=1, seed=seed)
        if version in ['standard', 'plus', 'rand']:
            self.set_version(version)

    # ------------------------------------------------------------------------------------------ small helpers
    def get_logits(self, x):
        return self.model(x)

    def get_seed(self):
        return time.time() if self.seed is None else self.seed

    def _say(self, msg):
        if self.verbose:
            self.logger.log(msg)

    def _warn(self, msg):
        self.logger.log(f'Warning: {msg}')

    def _predict_batches(self, x_all, bs):
        """arg-max predictions of the model over x_all in chunks of bs (on self.device), returned on x_all's device."""
        preds = []
        for s in range(0, x_all.shape[0], bs):
            preds.append(self.get_logits(x_all[s:s + bs].clone().to(self.device)).max(dim=1)[1].to(x_all.device))
        return torch.cat(preds) if preds else torch.empty(0, dtype=torch.long, device=x_all.device)

    def _defense_checks(self, x, y):
        """The reference's pre-flight warnings (autoattack/checks.py) on the first batch: randomised defence (5 passes
        must agree), outputs that look like probabilities, autograd activity inside the forward, enough classes for
        the (targeted) DLR loss.  Returns the number of classes."""
        with torch.no_grad():
            if self.version != 'rand':
                outs = [self.get_logits(x) for _ in range(5)]
                hits = [(o.max(1)[1] == y).sum().item() for o in outs]
                unit = [o / (o.reshape(o.shape[0], -1).norm(dim=1, keepdim=True) + 1e-10) for o in outs]
                spread = max(float((unit[a] - unit[b]).reshape(x.shape[0], -1).norm(dim=1).max())
                             for a in range(5) for b in range(a + 1, 5))
                if len(set(hits)) > 1 or spread > 1e-4:
                    self._warn('it seems to be a randomized defense! Please use version="rand".')
            out = self.get_logits(x)
            if out.max() < 1. + 1e-5 and out.min() > -1e-5 and ((out.sum(-1) - 1.).abs() < 1e-5).all():
                self._warn('it seems that the output is a probability distribution, please be sure that the logits are used!')
            n_cls = out.shape[-1]
        seen = {"n": 0}

        def tracer(frame, event, arg):
            if event == 'call' and frame.f_code.co_name in ('grad', 'backward', '_make_grads'):
                seen["n"] += 1
        sys.settrace(tracer)
        try:
            self.model(x)
        finally:
            sys.settrace(None)
        if seen["n"]:
            self._warn('it seems to be a dynamic defense! The evaluation with AutoAttack might be insufficient.')
        dlr_like = 'apgd-dlr' in self.attacks_to_run or 'apgd-t' in self.attacks_to_run
        if dlr_like and n_cls <= 2:
            self._warn(f'with only {n_cls} classes it is not possible to use the DLR loss!')
        elif dlr_like and n_cls == 3:
            self._warn(f'with only {n_cls} classes it is not possible to use the targeted DLR loss!')
        elif 'apgd-t' in self.attacks_to_run and self.apgd_targeted.n_target_classes + 1 > n_cls:
            self._warn(f'it seems that more target classes ({self.apgd_targeted.n_target_classes}) than possible '
                       f'({n_cls - 1}) are used in APGD-T!')
        return n_cls

    def _attack_batch(self, attack, x, y):
        if attack == 'apgd-ce' or attack == 'apgd-dlr':
            self.apgd.loss = attack[5:]
            self.apgd.seed = self.get_seed()
            return self.apgd.perturb(x, y)
        if attack == 'apgd-t':
            self.apgd_targeted.seed = self.get_seed()
            return self.apgd_targeted.perturb(x, y)
        if attack == 'square' and isinstance(self.square, SquareAttack):
            self.square.seed = self.get_seed()
            return self.square.perturb(x, y)
        if attack in _KNOWN:
            raise NotImplementedError(f"attack '{attack}' is not part of the native path (built: {', '.join(_BUILT)})")
        raise ValueError('Attack not supported')

    # ------------------------------------------------------------------------------------------ evaluation
    def run_standard_evaluation(self, x_orig, y_orig, bs=250, return_labels=False, state_path=None):
        if state_path is not None and Path(state_path).exists():
            state = EvaluationState.from_disk(state_path)
            if set(self.attacks_to_run) != state.attacks_to_run:
                raise ValueError("The state was created with a different set of attacks to run. "
                                 "You are probably using the wrong state file.")
            self._say("Restored state from {}".format(state_path))
            self._say("Since the state has been restored, **only** the adversarial examples from the current run "
                      "are going to be returned.")
        else:
            state = EvaluationState(set(self.attacks_to_run), path=state_path)
            state.to_disk()
            if state_path is not None:
                self._say("Created state in {}".format(state_path))
        todo = [a for a in self.attacks_to_run if a not in state.run_attacks]
        missing = [a for a in todo if a in _KNOWN and (a not in _BUILT or (a == 'square' and not isinstance(
            self.square, SquareAttack)))]
        if missing:       # fail before hours of attack time are spent, not when the unbuilt attack's turn comes
            raise NotImplementedError(f"attack(s) {missing} are not part of the native path (built: {', '.join(_BUILT)}; "
                                      f"square is L-inf only)")
        self._say('using {} version including {}.'.format(self.version, ', '.join(todo)))
        if state.run_attacks:
            self._say('{} was/were already run.'.format(', '.join(state.run_attacks)))
        self._defense_checks(x_orig[:bs].to(self.device), y_orig[:bs].to(self.device))

        n = x_orig.shape[0]
        with torch.no_grad():
            if state.robust_flags is None:                                # clean pass
                y_adv = self._predict_batches(x_orig, bs).to(y_orig.dtype)
                robust = y_orig.eq(y_adv.to(y_orig.device)).to(x_orig.device)
                state.robust_flags = robust
                state.clean_accuracy = robust.sum().item() / n
                acc_by_attack = {'clean': state.clean_accuracy}
                self._say('initial accuracy: {:.2%}'.format(state.clean_accuracy))
            else:
                robust = state.robust_flags.to(x_orig.device)
                y_adv = torch.empty_like(y_orig)
                acc_by_attack = {'clean': state.clean_accuracy}
                self._say('initial clean accuracy: {:.2%}'.format(state.clean_accuracy))
                self._say('robust accuracy at the time of restoring the state: {:.2%}'.format(robust.sum().item() / n))
            x_adv = x_orig.clone().detach()
            t0 = time.time()
            for attack in todo:
                alive = torch.nonzero(robust, as_tuple=False).flatten()      # indices of the still-robust points
                if alive.numel() == 0:
                    break
                n_chunks = int(np.ceil(alive.numel() / bs))
                for ci in range(n_chunks):
                    idx = alive[ci * bs:(ci + 1) * bs]
                    x = x_orig[idx].clone().to(self.device)
                    y = y_orig[idx].clone().to(self.device)
                    with torch.enable_grad():
                        adv = self._attack_batch(attack, x, y)
                    pred = self.get_logits(adv).max(dim=1)[1]
                    fooled = ~y.eq(pred).to(robust.device)
                    robust[idx[fooled]] = False
                    state.robust_flags = robust
                    x_adv[idx[fooled]] = adv[fooled].detach().to(x_adv.device)
                    y_adv[idx[fooled]] = pred[fooled].detach().to(y_adv.device)
                    self._say('{} - {}/{} - {} out of {} successfully perturbed'.format(
                        attack, ci + 1, n_chunks, int(fooled.sum()), x.shape[0]))
                acc_by_attack[attack] = robust.sum().item() / n
                state.add_run_attack(attack)
                self._say('robust accuracy after {}: {:.2%} (total time {:.1f} s)'.format(
                    attack.upper(), acc_by_attack[attack], time.time() - t0))
            # checks.py:73-87: Square is the weakest attack of the ensemble - if it alone lowers the robust accuracy the
            # white-box attacks are probably failing (gradient masking)
            if 'square' in acc_by_attack and len(acc_by_attack) > 2:
                floor = min(v for k, v in acc_by_attack.items() if k != 'square')
                if acc_by_attack['square'] < floor - .002:
                    self._warn('Square Attack has decreased the robust accuracy of {:.2%}. This might indicate that the '
                               'robustness evaluation using AutoAttack is unreliable. Consider running Square Attack '
                               'with more iterations and restarts or an adaptive attack.'.format(
                                   floor - acc_by_attack['square']))
            state.to_disk(force=True)
            if self.verbose:
                d = (x_adv - x_orig).reshape(n, -1)
                res = d.abs().max(1)[0] if self.norm == 'Linf' else (d ** 2).sum(-1).sqrt() if self.norm == 'L2' \
                    else d.abs().sum(-1)
                self.logger.log('max {} perturbation: {:.5f}, nan in tensor: {}, max: {:.5f}, min: {:.5f}'.format(
                    self.norm, res.max(), (x_adv != x_adv).sum(), x_adv.max(), x_adv.min()))
                self.logger.log('robust accuracy: {:.2%}'.format(robust.sum().item() / n))
        return (x_adv, y_adv) if return_labels else x_adv

    def clean_accuracy(self, x_orig, y_orig, bs=250):
        with torch.no_grad():
            hits = (self._predict_batches(x_orig, bs).to(y_orig.device) == y_orig).float().sum().item()
        if self.verbose:
            print('clean accuracy: {:.2%}'.format(hits / x_orig.shape[0]))
        return hits / x_orig.shape[0]

    def run_standard_evaluation_individual(self, x_orig, y_orig, bs=250, return_labels=False):
        """Each attack on its own, from the clean points (autoattack.py:268-291)."""
        self._say('using {} version including {}'.format(self.version, ', '.join(self.attacks_to_run)))
        plan, was_verbose = self.attacks_to_run, self.verbose
        self.verbose = False
        out = {}
        try:
            for attack in plan:
                t0 = time.time()
                self.attacks_to_run = [attack]
                x_adv, y_adv = self.run_standard_evaluation(x_orig, y_orig, bs=bs, return_labels=True)
                out[attack] = (x_adv, y_adv) if return_labels else x_adv
                if was_verbose:
                    acc = self.clean_accuracy(x_adv, y_orig, bs=bs)
                    self.logger.log('robust accuracy by {} \t {:.2%} \t (time attack: {:.1f} s)'.format(
                        attack.upper(), acc, time.time() - t0))
        finally:
            self.attacks_to_run, self.verbose = plan, was_verbose
        return out

    def set_version(self, version='standard'):
        """Attack lists / budgets of the named versions (autoattack.py:293-330)."""
        if self.verbose:
            print('setting parameters for {} version'.format(version))
        small_norm = self.norm in ['Linf', 'L2']
        if version == 'standard':
            self.attacks_to_run = ['apgd-ce', 'apgd-t', 'fab-t', 'square']
            self.apgd.n_restarts = 1 if small_norm else 5
            self.apgd_targeted.n_target_classes = 9 if small_norm else 5
            self.apgd_targeted.n_restarts = 1
            self.fab.n_restarts, self.fab.n_target_classes, self.square.n_queries = 1, 9, 5000
        elif version == 'plus':
            self.attacks_to_run = ['apgd-ce', 'apgd-dlr', 'fab', 'square', 'apgd-t', 'fab-t']
            self.apgd.n_restarts, self.fab.n_restarts, self.apgd_targeted.n_restarts = 5, 5, 1
            self.fab.n_target_classes, self.apgd_targeted.n_target_classes, self.square.n_queries = 9, 9, 5000
            if not small_norm:
                print('"{}" version is used with {} norm: please check'.format(version, self.norm))
        elif version == 'rand':
            raise NotImplementedError("version='rand' needs EOT (eot_iter=20), which the native APGD does not run")
        else:
            raise ValueError(f"unknown version {version}")
