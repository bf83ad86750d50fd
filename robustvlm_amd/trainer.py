"""Outer FARE / TeCoA training step on the native engine.

Mirrors the step semantics of ``train_one_epoch`` (train/adversarial_training_clip.py:289-427):

    e0 = model_orig(x)                        (frozen copy, no grad)            :296-297
    x_adv = pgd | apgd (model in eval mode)                                     :305-336
    emb_clean = model(x) ; emb_adv = model(x_adv)      (train mode)             :340,349
    loss_clean = compute_loss(loss_clean, emb_clean, e0, T=None)  if cw > 0     :341-347
    loss = compute_loss(loss, emb_adv, e0 | emb_clean.detach() (trades), T)     :352-359
    loss_total = cw * loss_clean + (1 - cw) * loss                              :360
    loss_total.backward(); optimizer.step(); zero_grad();                       :361-364
    step_total += 1; scheduler(step_total)   (sets the NEXT step's LR)          :365-366
    cos-sim-clean / cos-sim / acc / racc                (logging, no grad)      :368-387
    every eval_freq steps: acc / racc / cos-sim under a 50-step supervised APGD :389-424

with AdamW (lr 1e-5, wd 1e-4, :196-197) and open_clip's cosine_lr with linear warm-up (:211).
The reference wraps the model in single-process nn.DataParallel (:184-191), whose replica gradients are reduced
inside ``backward``; here every rank owns one GPU and its shard of the batch, the attack needs no communication, and
the gradient all-reduce (RCCL through torch.distributed) runs in ``n_buckets`` contiguous slices of ONE flat fp32
buffer, each launched as soon as the backward stages that fill it are done, so that it overlaps the remaining stages
(rvlm_vit_backward_params_stages); the 1/world_size scaling is fused into the AdamW kernel.
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist

from . import _lib as L
from .clip_model import ClipVisionModel, ComputeLossWrapper, compute_loss
from .config import state_dict_shapes, backward_stage_keys, parameter_order
from .dist import allreduce_sum_span
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


def bucket_plan(n_stages: int, n_buckets: int):
    """[(stage_begin, stage_end)] covering [0, n_stages) in order, sizes as even as possible."""
    n_buckets = max(1, min(int(n_buckets), n_stages))
    base, rem = divmod(n_stages, n_buckets)
    out, lo = [], 0
    for b in range(n_buckets):
        hi = lo + base + (1 if b < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


class FlatParams:
    """All parameters of the vision tower as ONE flat fp32 buffer (+ per-key views), so that the optimizer is a single
    launch over 304 M elements and a gradient bucket is a contiguous slice.  The buffer is laid out in BACKWARD-STAGE
    order (head, block L-1 ... block 0, embeddings): the keys of stages [a, b) occupy ``stage_span(a, b)``."""

    def __init__(self, cfg, state_dict: dict, device):
        self.shapes = state_dict_shapes(cfg)
        self.stage_keys = backward_stage_keys(cfg)
        self.offsets, self.stage_offsets, n = {}, [0], 0
        for keys in self.stage_keys:
            for k in keys:
                cnt = 1
                for d in self.shapes[k]:
                    cnt *= d
                self.offsets[k] = (n, cnt)
                n = (n + cnt + 3) // 4 * 4        # keep every tensor 16-byte aligned
            self.stage_offsets.append(n)
        assert set(self.offsets) == set(self.shapes)
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + c].view(self.shapes[k]) for k, (o, c) in self.offsets.items()}
        if state_dict is not None:
            for k, v in self.views.items():
                v.copy_(state_dict[k].detach().to(device=device, dtype=torch.float32))

    def like(self):
        other = FlatParams.__new__(FlatParams)
        other.shapes, other.offsets, other.numel = self.shapes, self.offsets, self.numel
        other.stage_keys, other.stage_offsets = self.stage_keys, self.stage_offsets
        other.flat = torch.zeros_like(self.flat)
        other.views = {k: other.flat[o:o + c].view(self.shapes[k]) for k, (o, c) in self.offsets.items()}
        return other

    def stage_span(self, a: int, b: int):
        return self.stage_offsets[a], self.stage_offsets[b]

    def state_dict(self):
        return {k: self.views[k].clone() for k in self.shapes}          # visual.state_dict() key order


class AdversarialTrainer:
    """One process per GPU.  ``state_dict`` = open_clip ``visual.state_dict()`` of the model to fine-tune;
    the frozen ``model_orig`` copy is created from the same weights (…clip.py:95-97,183-186)."""

    def __init__(self, cfg, state_dict, batch_size, precision="bf16", lr=1e-5, wd=1e-4, warmup=1400,
                 steps=20000, loss="l2", inner_loss="l2", attack="pgd", norm="linf", eps=4 / 255,
                 iterations_adv=10, stepsize_adv=1 / 255, output_normalize=False, clean_weight=0.0,
                 embedding_text_labels_norm=None, betas=(0.9, 0.999), adam_eps=1e-8, device=None,
                 loss_clean="l2", trades=False, n_buckets=4, metrics=True, process_group=None,
                 always_reduce=False):
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
        self.loss_clean = loss_clean                              # --loss_clean (…clip.py:343)
        self.trades = bool(trades)                                # --trades (…clip.py:352-358)
        self.eps, self.iterations_adv, self.stepsize_adv = eps, iterations_adv, stepsize_adv
        self.output_normalize, self.clean_weight = output_normalize, clean_weight
        self.T = embedding_text_labels_norm
        self.metrics = bool(metrics)
        self.step_total = 0
        # The reference builds the scheduler (…clip.py:211) but calls it only AFTER each optimizer step (:366, its one
        # call site): the first step of a fresh run uses the BASE learning rate AdamW was built with (:197), step s >= 2
        # uses cosine_lr(s - 1).  Pinned by tests/golden/train_step_tiny.npz (the reference's own train_one_epoch).
        self.cur_lr = float(lr)
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1
        # always_reduce: run the bucketed all-reduce path even in a one-rank group (exercises the RCCL calls, the work
        # handles and the stream ordering on a single GPU; the reduction itself is then the identity)
        self._reduce = self.world > 1 or (bool(always_reduce) and dist.is_available() and dist.is_initialized())
        self.buckets = bucket_plan(self.engine.n_stages, n_buckets if self._reduce else 1)
        # a backend without device collectives (gloo in the CPU/1-GPU tests) reduces through a pinned host copy
        self._device_collectives = self._reduce and dist.get_backend(self.pg) == "nccl"
        self._checked_global_batch = False

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

    def _allreduce_span(self, lo: int, hi: int):
        """Sum all-reduce of grads.flat[lo:hi] over the ranks; returns a waitable (or None when already done)."""
        return allreduce_sum_span(self.grads.flat, lo, hi, self.pg, self._device_collectives)

    def _loss_backward(self, x, targets, e_ref, loss_str, T, weight, accumulate, reduce_grads):
        """forward (activations kept for wgrad) + loss + weight gradients scaled by ``weight``; with ``reduce_grads``
        the gradient buckets are all-reduced while the later backward stages still run."""
        emb = self.engine.forward(x, None, self.output_normalize, save=2)
        e = emb.detach().requires_grad_(True)
        loss = compute_loss(loss_str, e, targets, e_ref, 100., T)               # reduction='mean'
        (d_emb,) = torch.autograd.grad(loss, e)
        d_emb = d_emb * weight
        if not reduce_grads:
            self.engine.backward_params(d_emb, self.grads.views, accumulate=accumulate)
            return loss.detach(), emb
        works = []
        for a, b in self.buckets:
            self.engine.backward_params(d_emb, self.grads.views, accumulate=accumulate, stages=(a, b))
            works.append(self._allreduce_span(*self.params.stage_span(a, b)))   # async: overlaps the next stages
        for w in works:
            if w is not None:
                w.wait()                                                         # the compute stream waits, not the host
        return loss.detach(), emb

    def _verify_pending_global_batch(self):
        pend, self._pending_global_batch = getattr(self, "_pending_global_batch", None), None
        if pend is not None and int(pend[0].item()) != pend[1]:
            raise ValueError(f"global_batch={pend[1]} but the ranks held {int(pend[0].item())} images in that step "
                             f"(its gradients were mis-scaled by {pend[1] / max(int(pend[0].item()), 1):.4g})")

    def _shard_weight(self, n_local: int, global_batch: int | None = None) -> float:
        """The reference's loss is the mean over the GLOBAL batch (DataParallel gathers the outputs first, …clip.py:
        184-191).  Here every rank takes the mean over its shard and AdamW divides the summed gradients by world_size;
        the factor n_local * world / n_global makes that exact for uneven shards too (1.0 for equal ones).  A caller
        that knows the global batch size passes it: no collective and no host sync in front of the backward then."""
        if self.world == 1:
            return 1.0
        if global_batch is not None:
            # A wrong value (the last partial batch of an epoch passed with the nominal size, uneven shards) would silently
            # mis-scale every rank's gradients, so EVERY step that is given one checks it against the all-reduced shard
            # sizes - without a host sync in front of the backward: the tiny all-reduce is enqueued now, its result is
            # read at the start of the NEXT step (long complete by then); only the first step waits for it.  Every rank
            # must pass global_batch (or omit it) consistently: the check is a collective.
            self._verify_pending_global_batch()
            n = torch.tensor([float(n_local)], dtype=torch.float64, device=self.device if self._device_collectives else None)
            dist.all_reduce(n, op=dist.ReduceOp.SUM, group=self.pg)
            self._pending_global_batch = (n, int(global_batch))
            if not self._checked_global_batch:
                self._verify_pending_global_batch()
                self._checked_global_batch = True
            return n_local * self.world / float(global_batch)
        n = torch.tensor([float(n_local)], dtype=torch.float64)
        if self._device_collectives:
            n = n.to(self.device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=self.pg)
        return n_local * self.world / float(n.item())

    @torch.no_grad()
    def _cos_mean(self, a, b):
        per = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        mean = torch.empty(1, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            L.check(self.lib.rvlm_cosine_rows(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], per.data_ptr(),
                                              mean.data_ptr(), L.stream_ptr()), "rvlm_cosine_rows")
        return mean.reshape(())

    @torch.no_grad()
    def _acc(self, emb, targets, normalize):
        """compute_acc(emb[_norm] @ T, targets) (…clip.py:374-379,488-492) on device kernels; percent."""
        B, D = emb.shape
        T = self.T.detach().to(device=emb.device, dtype=torch.float32).contiguous()
        tg = targets.detach().to(torch.int64).contiguous()
        e = emb.contiguous()
        pred = torch.empty(B, dtype=torch.uint8, device=emb.device)
        logits = torch.empty(B, T.shape[1], dtype=torch.float32, device=emb.device)
        with torch.cuda.device(emb.device):
            if normalize:
                en, inv = torch.empty_like(e), torch.empty(B, dtype=torch.float32, device=emb.device)
                L.check(self.lib.rvlm_l2_normalize_rows(e.data_ptr(), B, D, en.data_ptr(), inv.data_ptr(), L.stream_ptr()))
                e = en
            L.check(self.lib.rvlm_head_logits(e.data_ptr(), T.data_ptr(), B, D, T.shape[1], 1.0, logits.data_ptr(),
                                              L.stream_ptr()), "rvlm_head_logits")
            L.check(self.lib.rvlm_argmax_eq(logits.data_ptr(), tg.data_ptr(), B, T.shape[1], pred.data_ptr(),
                                            L.stream_ptr()), "rvlm_argmax_eq")
        return (pred.sum() / B).item() * 100

    def _global_means(self, out: dict, n_local: int) -> dict:
        """Logging values over the GLOBAL batch, as the reference's single-process DataParallel reports them (…clip.py:
        368-387 on gathered outputs): every rank contributes n_local x its shard mean, one small all-reduce."""
        keys = [k for k in ("loss", "loss_clean", "loss_total", "cos_sim_clean", "cos_sim", "acc", "racc") if out.get(k) is not None]
        # stacked on the device: ONE host sync (after the all-reduce) instead of one float() per metric
        vals = [out[k].detach().to(device=self.device, dtype=torch.float64).reshape(()) if isinstance(out[k], torch.Tensor)
                else torch.tensor(float(out[k]), dtype=torch.float64, device=self.device) for k in keys]
        vec = torch.cat([torch.stack(vals) * n_local, torch.tensor([float(n_local)], dtype=torch.float64, device=self.device)])
        if not self._device_collectives:
            vec = vec.cpu()
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.pg)
        vec = vec.cpu()
        res = dict(out)
        for i, k in enumerate(keys):
            res[k] = float(vec[i] / vec[-1])
        return res

    def train_step(self, data, targets, data_adv=None, global_batch=None, global_metrics=False):
        """One optimizer step on this rank's shard.  Returns dict(loss, loss_clean, loss_total, lr) and, with
        ``metrics`` on, the reference's logging values cos_sim_clean, cos_sim, acc, racc (acc / racc None unless
        ``targets`` are labels and a text head was given).  ``data_adv`` (optional) bypasses the attack with
        precomputed adversarial images (tests).  ``global_batch`` (optional): the number of images all ranks hold in
        this step, when the caller knows it (equal shards: world * len(data)) - skips the size all-reduce.  The
        returned loss / metrics are this rank's SHARD values; ``global_metrics=True`` (one extra 8-number all-reduce and
        a host sync, for logging steps) returns them over the global batch like the reference's DataParallel logs."""
        with torch.no_grad():
            e0 = self.model_orig(data, self.output_normalize)                   # …clip.py:296-297
        if data_adv is None:
            data_adv = self._attack(data, targets, e0)
        cw = self.clean_weight
        wshard = self._shard_weight(data.shape[0], global_batch)
        dp = self._reduce
        loss_clean = torch.zeros((), device=self.device)
        emb_clean = None
        accumulate = False
        if cw > 0.:                                                              # …clip.py:341-347
            loss_clean, emb_clean = self._loss_backward(data, targets, e0, self.loss_clean, None, cw * wshard, False, False)
            accumulate = True
        elif self.metrics or self.trades:
            # the reference runs the clean forward in every step (…clip.py:340); its embedding feeds the logging
            # metrics and the TRADES target only, so no activations are kept
            emb_clean = self.engine.forward(data, None, self.output_normalize, save=0)
        e_ref = emb_clean.detach().clone() if self.trades else e0              # …clip.py:352-358
        loss, emb_adv = self._loss_backward(data_adv, targets, e_ref, self.loss, self.T, (1.0 - cw) * wshard,
                                            accumulate, dp)                      # …clip.py:349-361
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
        out = dict(loss=loss, loss_clean=loss_clean, loss_total=cw * loss_clean + (1 - cw) * loss, lr=lr_used)
        if self.metrics:                                                         # …clip.py:368-387
            out["cos_sim_clean"] = self._cos_mean(emb_clean, e0)
            out["cos_sim"] = self._cos_mean(emb_adv, e0)
            is_cls = isinstance(targets, torch.Tensor) and self.T is not None
            out["racc"] = self._acc(emb_adv, targets, normalize=False) if is_cls else None   # logits_adv: NOT normalised
            out["acc"] = self._acc(emb_clean, targets, normalize=True) if is_cls else None
        if global_metrics and self.world > 1:
            out = self._global_means(out, data.shape[0])
        return out

    def profile_allreduce(self, data, reps: int = 3) -> dict:
        """Self-diagnosis of the data-parallel step (bench.py --mode train --gpus N): every gradient bucket's RCCL
        all-reduce timed ALONE (compute stream idle), and one weight-gradient backward timed without / with the bucketed
        reduction launched under its later stages - the exposed time is what the overlap did not hide.  Leaves the
        parameters untouched (no optimizer step); the gradient buffer is scratch afterwards."""
        if not self._reduce:
            return {"note": "no gradient all-reduce in this configuration (one rank, always_reduce off)"}
        with torch.no_grad():
            e0 = self.model_orig(data, self.output_normalize)
        ev = lambda: torch.cuda.Event(enable_timing=True)        # noqa: E731
        spans = [self.params.stage_span(a, b) for a, b in self.buckets]
        alone = []
        for lo, hi in spans:
            best = None
            for _ in range(reps):
                torch.cuda.synchronize(self.device)
                e0_, e1_ = ev(), ev()
                e0_.record()
                w = self._allreduce_span(lo, hi)
                if w is not None:
                    w.wait()
                e1_.record()
                torch.cuda.synchronize(self.device)
                t = e0_.elapsed_time(e1_)
                best = t if best is None else min(best, t)
            alone.append(best)

        def backward_ms(reduce_grads):
            best = None
            for _ in range(reps):
                torch.cuda.synchronize(self.device)
                a, b = ev(), ev()
                a.record()
                self._loss_backward(data, None, e0 + 0.01, "l2", None, 1.0, False, reduce_grads)
                b.record()
                torch.cuda.synchronize(self.device)
                t = a.elapsed_time(b)
                best = t if best is None else min(best, t)
            return best
        t_plain, t_red = backward_ms(False), backward_ms(True)
        total = sum(alone)
        exposed = max(t_red - t_plain, 0.0)
        return {"n_buckets": len(spans), "bucket_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi in spans],
                "bucket_allreduce_ms_alone": [round(t, 3) for t in alone], "allreduce_ms_alone_total": round(total, 3),
                "fwd_bwd_ms_without_reduce": round(t_plain, 3), "fwd_bwd_ms_with_bucketed_reduce": round(t_red, 3),
                "exposed_ms": round(exposed, 3),
                "overlapped_fraction": round(1.0 - exposed / total, 4) if total > 0 else None,
                "backend": "nccl (RCCL)" if self._device_collectives else "host copy (gloo)", "world": self.world}

    def eval_step(self, data_eval, targets_eval):
        """The periodic validation of …clip.py:389-424: acc / racc against a supervised 50-step APGD (CE on the
        zero-shot head, ``initial_stepsize = 0.05 * eps`` when clean_weight > 0) and the clean-vs-adversarial
        cosine similarity.  Runs on the model being trained, in eval mode."""
        assert self.T is not None, "eval_step needs embedding_text_labels_norm"
        was_training = self.model.training
        self.model.eval()
        wrap = ComputeLossWrapper(None, self.T, "none", "ce", 100.)
        adv = apgd_train(self.model, data_eval, targets_eval, self.norm, self.eps, n_iter=50, loss_fn=wrap,
                         initial_stepsize=0.05 * self.eps if self.clean_weight > 0 else None, verbose=False)
        with torch.no_grad():
            e_adv = self.model(adv, True)
            e_cln = self.model(data_eval, True)
            logs = {"eval/racc": self._acc(e_adv, targets_eval, normalize=False),
                    "eval/acc": self._acc(e_cln, targets_eval, normalize=False),
                    "eval/cos-sim": float(self._cos_mean(e_adv, e_cln))}
        self.model.train(was_training)
        return logs

    def state_dict(self):
        """open_clip ``visual.state_dict()`` of the fine-tuned tower (what …clip.py:239,470 saves).  Checkpoint time is also where
        the LAST step's deferred ``global_batch`` check is settled (it is otherwise read at the start of the next step: a wrong
        value passed with an epoch's final partial batch would never be reported)."""
        self._verify_pending_global_batch()
        return self.params.state_dict()

    # -- optimizer state: the file torch.optim.AdamW.state_dict() would write for visual.parameters() ---------------
    def optimizer_state_dict(self):
        """``torch.optim.AdamW(visual.parameters()).state_dict()`` layout (…clip.py:196-197,240,471): 'state' keyed by the
        positional index of ``visual.parameters()``, one param group - a reference run can resume from it and vice
        versa.  'param_names' (index -> key) is added for readers without the module (torch ignores it)."""
        order = parameter_order(self.cfg)

        def piece(flat, k):
            o, c = self.params.offsets[k]
            return flat[o:o + c].view(self.params.shapes[k]).clone()
        state = {i: {"step": torch.tensor(float(self.step_total)), "exp_avg": piece(self.exp_avg, k),
                     "exp_avg_sq": piece(self.exp_avg_sq, k)} for i, k in enumerate(order)}
        group = {"lr": self.cur_lr, "betas": tuple(self.betas), "eps": self.adam_eps, "weight_decay": self.wd,
                 "amsgrad": False, "foreach": None, "maximize": False, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group], "param_names": order}

    def load_optimizer_state_dict(self, sd, start_step=None):
        """Resume (…clip.py:207-208,219): restores the moments and the step counter; the LR schedule is a pure
        function of the step.  Accepts torch's AdamW layout (from this trainer or from a reference run) and the
        name-keyed layout round 1 of this package wrote."""
        if "state" in sd and "param_groups" in sd:
            order = parameter_order(self.cfg)
            if len(sd["state"]) not in (0, len(order)):
                raise ValueError(f"optimizer state has {len(sd['state'])} entries, the tower {len(order)} parameters")
            step = 0
            for i, k in enumerate(order):
                if i not in sd["state"]:
                    continue
                st, (o, c) = sd["state"][i], self.params.offsets[k]
                if tuple(st["exp_avg"].shape) != tuple(self.params.shapes[k]):
                    raise ValueError(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter {k} "
                                     f"{tuple(self.params.shapes[k])}: not visual.parameters() order")
                self.exp_avg[o:o + c].copy_(st["exp_avg"].reshape(-1).to(self.device))
                self.exp_avg_sq[o:o + c].copy_(st["exp_avg_sq"].reshape(-1).to(self.device))
                step = int(float(st["step"]))
            file_lr = sd["param_groups"][0].get("lr") if sd["param_groups"] else None
        else:
            for k, (o, c) in self.params.offsets.items():
                self.exp_avg[o:o + c].copy_(sd["exp_avg"][k].reshape(-1).to(self.device))
                self.exp_avg_sq[o:o + c].copy_(sd["exp_avg_sq"][k].reshape(-1).to(self.device))
            step = int(sd["step"])
            file_lr = None
        self.step_total = int(step if start_step is None else start_step)
        # optimizer.load_state_dict restores the param group's lr (…clip.py:207-208) = what the scheduler left behind
        # when the file was written; the next scheduler call comes after the first resumed step
        self.cur_lr = float(file_lr) if file_lr is not None else cosine_lr_value(self.step_total, self.lr, self.warmup,
                                                                                 self.steps)

    def load_state_dict(self, state_dict):
        """Resume the model weights (…clip.py:98-103): both engines' GEMM copies are refreshed; ``model_orig`` keeps the
        ORIGINAL weights it was created with unless ``load_orig_state_dict`` is called."""
        for k, v in self.params.views.items():
            v.copy_(state_dict[k].detach().to(device=self.device, dtype=torch.float32))
        self.engine.load_state_dict(self.params.views)

    def close(self):
        try:
            self._verify_pending_global_batch()      # (the last step's deferred check; raised after the engines are released)
        finally:
            self.engine.close()
            self.engine_orig.close()
