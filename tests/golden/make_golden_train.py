#!/usr/bin/env python3
"""Golden vectors for the optimizer step (SURVEY.md 8(f) rank 1 / the second half of row a10), produced by RUNNING THE
REFERENCE'S OWN ``train_one_epoch`` in the build container.

Run once from the repo root:  python tests/golden/make_golden_train.py
Needs /root/reference (read-only), so it never runs on the GPU box; only its output tests/golden/train_step_tiny.npz
(inputs + expected outputs) is committed.

What is executed from the reference (nothing is copied into this repo):
  * ``train_one_epoch``, ``compute_loss``, ``l2``, ``ce``, ``ComputeLossWrapper``, ``compute_acc`` of
    train/adversarial_training_clip.py (:260-274, :276-486, :488-528) - the module itself needs torchvision / open_clip /
    wandb to import, so the definitions are pulled out of the file with ``ast`` at run time and exec'd as they are;
  * ``AverageMeter`` of train/utils.py (:33-55), same way (that module imports wandb);
  * ``train.pgd_train.pgd`` and ``train.apgd_train.apgd_train`` - imported from /root/reference;
  * ``torch.optim.AdamW`` - the reference's optimizer (:196-197), used directly.
What the harness supplies (the things ``train_one_epoch`` takes as PARAMETERS or reads as globals):
  * ``model`` / ``model_orig``: the tiny seeded ViT of oracle/vit_ref.py (pinned against HF transformers) wrapped as an
    nn.Module whose ``.model.parameters()`` are the open_clip-keyed tensors;
  * ``scheduler``: open_clip ``training.scheduler.cosine_lr`` (third party, open-clip-torch==2.19.0, requirements.txt:91;
    imported at :18, built at :211) restated from its published source: ``assign_learning_rate`` of
    ``base_lr * (step + 1) / warmup_length`` below ``warmup_length``, else ``0.5 * (1 + cos(pi * e / es)) * base_lr``;
  * a list of batches as ``dataloader`` / ``dataloader_eval``; ``args`` = a namespace with the CLI fields the function
    reads; ``wandb`` = a recorder whose ``log`` keeps the reference's ``log_data`` and a snapshot of the parameters;
    ``Tensor.cuda`` / ``torch.cuda.empty_cache`` are neutralised (CPU container).
``pgd`` / ``apgd`` are wrapped only to RECORD the adversarial batch they return (the random start comes from torch's
global generator, which a device implementation cannot replay).

Recorded per case and step: the batch, the adversarial batch, the learning rate the optimizer step USED and the one the
scheduler left behind, loss / loss-total / cos-sim-clean / cos-sim / acc / racc, every parameter after the step; and
the periodic validation (``eval/*``) that the reference runs at the first step.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import vit_ref  # noqa: E402
from oracle.vit_ref import init_weights, vit_forward, normalize_pixels  # noqa: E402
from train.pgd_train import pgd as ref_pgd  # noqa: E402
from train.apgd_train import apgd_train as ref_apgd  # noqa: E402

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def _extract(path, want):
    tree = ast.parse(open(path).read())
    nodes = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in nodes} == set(want), {n.name for n in nodes}
    return nodes, path


class TinyClipVision(torch.nn.Module):
    """ClipVisionModel-shaped module (…clip.py:246-257) over the oracle's ViT: ``.model`` holds the parameters."""

    def __init__(self, cfg, weights):
        super().__init__()
        self.cfg, self.keys = cfg, list(weights)
        self.model = torch.nn.ParameterList([torch.nn.Parameter(weights[k].clone()) for k in self.keys])

    def named(self):
        return dict(zip(self.keys, self.model))

    def forward(self, vision, output_normalize):
        e = vit_forward(self.cfg, self.named(), normalize_pixels(vision))
        return torch.nn.functional.normalize(e, dim=-1) if output_normalize else e


def cosine_lr(optimizer, base_lr, warmup_length, steps):
    """open_clip training.scheduler.cosine_lr [3p], restated (see the module docstring)."""
    def _lr_adjuster(step):
        if step < warmup_length:
            lr = base_lr * (step + 1) / warmup_length
        else:
            e, es = step - warmup_length, steps - warmup_length
            lr = 0.5 * (1 + np.cos(np.pi * e / es)) * base_lr
        for g in optimizer.param_groups:
            g["lr"] = lr
        return lr
    return _lr_adjuster


class WandbRecorder:
    def __init__(self):
        self.rows, self.model = [], None

    def log(self, data):
        row = dict(data)
        row["_weights"] = {k: v.detach().clone() for k, v in self.model.named().items()}
        self.rows.append(row)


def build_namespace(rec, adv_log):
    nodes, path = _extract(os.path.join(REF, "train", "adversarial_training_clip.py"),
                           {"train_one_epoch", "compute_loss", "l2", "ce", "ComputeLossWrapper", "compute_acc"})
    mnodes, mpath = _extract(os.path.join(REF, "train", "utils.py"), {"AverageMeter"})

    def pgd(**kw):
        out = ref_pgd(**kw)
        adv_log.append(out.detach().clone())
        return out

    def apgd(**kw):
        out = ref_apgd(**kw)
        adv_log.append(out.detach().clone())
        return out

    import time
    ns = {"torch": torch, "F": torch.nn.functional, "time": time, "os": os, "pgd": pgd, "apgd": apgd, "wandb": rec,
          "unwrap_model": lambda m: m}
    exec(compile(ast.Module(body=mnodes, type_ignores=[]), mpath, "exec"), ns)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return ns


CASES = {
    # FARE (README.md:282): pgd, l2 / l2, embeddings not normalised
    "fare_pgd": dict(attack="pgd", loss="l2", inner_loss="l2", loss_clean="l2", clean_weight=0.0, trades=False,
                     output_normalize=False),
    # TeCoA (README.md:277): pgd, ce / ce, normalised embeddings
    "tecoa_pgd": dict(attack="pgd", loss="ce", inner_loss="ce", loss_clean="l2", clean_weight=0.0, trades=False,
                      output_normalize=True),
    # clean term, no attack (data_adv = data, …clip.py:334-335)
    "none_cw": dict(attack="none", loss="ce", inner_loss="ce", loss_clean="l2", clean_weight=0.5, trades=False,
                    output_normalize=True),
    # the CLI's default attack with TRADES targets and a clean term
    "apgd_trades_cw": dict(attack="apgd", loss="l2", inner_loss="l2", loss_clean="l2", clean_weight=0.3, trades=True,
                           output_normalize=True),
}
N_STEPS, B, C = 5, 4, 7
LR, WD, WARMUP, STEPS = 1e-3, 1e-2, 3, 6


def main():
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=21)
    g = torch.Generator().manual_seed(77)
    batches = [(torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g), torch.randint(0, C, (B,), generator=g))
               for _ in range(N_STEPS)]
    eval_batch = (torch.rand(6, 3, cfg.image_size, cfg.image_size, generator=g), torch.randint(0, C, (6,), generator=g))
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, C, generator=g), dim=0)
    eps, step = 4 / 255, 1 / 255

    arrs = dict(cfg=np.array([cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim]),
                weights_seed=np.int64(21), T=T.numpy(), eps=np.float64(eps), stepsize=np.float64(step),
                lr=np.float64(LR), wd=np.float64(WD), warmup=np.int64(WARMUP), steps=np.int64(STEPS),
                x=np.stack([b[0].numpy() for b in batches]), y=np.stack([b[1].numpy() for b in batches]),
                x_eval=eval_batch[0].numpy(), y_eval=eval_batch[1].numpy(), cases=np.array(sorted(CASES)))
    for k, v in w.items():
        arrs["w::" + k] = v.numpy()

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None

    for name, c in CASES.items():
        torch.manual_seed(1234)
        rec, adv_log = WandbRecorder(), []
        ns = build_namespace(rec, adv_log)
        model, model_orig = TinyClipVision(cfg, w), TinyClipVision(cfg, w)
        rec.model = model
        optimizer = torch.optim.AdamW(model.model.parameters(), lr=LR, weight_decay=WD)      # …clip.py:194-197
        scheduler = cosine_lr(optimizer, LR, WARMUP, STEPS)                                  # …clip.py:211
        lr_used = []
        inner_step = optimizer.step

        def step_and_note(*a, _inner=inner_step, _opt=optimizer, **k):
            lr_used.append(_opt.param_groups[0]["lr"])
            return _inner(*a, **k)
        optimizer.step = step_and_note
        args = types.SimpleNamespace(norm="linf", eps=eps, iterations_adv=10, stepsize_adv=step, eval_freq=1000,
                                     log_freq=1, save_checkpoints=False, steps=N_STEPS, total_epochs=1.0,
                                     output_dir="/nonexistent", **c)
        step_total = ns["train_one_epoch"](0, model=model, model_orig=model_orig, dataloader=batches,
                                           dataloader_eval=[eval_batch], optimizer=optimizer, scheduler=scheduler,
                                           embedding_text_labels_norm=T, normalize=None, args=args, epoch=0)
        assert step_total == N_STEPS and len(rec.rows) == N_STEPS and len(lr_used) == N_STEPS
        # the training batches' attacks; the first step also runs the 50-step validation attack AFTER its optimizer
        # step ((step_total - 1) % eval_freq == 0, …clip.py:389-424): order = train 1, eval, train 2, ...
        n_train_attacks = 0 if c["attack"] == "none" else N_STEPS
        assert len(adv_log) == n_train_attacks + 1
        arrs[f"{name}::x_adv_eval"] = adv_log.pop(1 if n_train_attacks else 0).numpy()
        if n_train_attacks:
            arrs[f"{name}::x_adv"] = np.stack([a.numpy() for a in adv_log])
        arrs[f"{name}::lr_used"] = np.array(lr_used, dtype=np.float64)
        arrs[f"{name}::lr_after"] = np.array([r["lr"] for r in rec.rows], dtype=np.float64)
        for key, col in (("loss", "loss"), ("loss_total", "loss-total"), ("cos_sim_clean", "cos-sim-clean"),
                         ("cos_sim", "cos-sim"), ("acc", "acc"), ("racc", "racc")):
            arrs[f"{name}::{key}"] = np.array([r[col] for r in rec.rows], dtype=np.float64)
        r0 = rec.rows[0]
        arrs[f"{name}::eval"] = np.array([r0["eval/acc"], r0["eval/racc"], float(r0["eval/cos-sim"])], dtype=np.float64)
        assert all("eval/acc" not in r for r in rec.rows[1:])
        # parameters in full after the steps that tell the schedules apart (fare_pgd: 1 = base LR, 2 = first warm-up
        # value, N = cosine part) and after the last step of every case; float64 sum and sum of squares of every
        # parameter tensor after EVERY step (keeps the fixture small)
        full = {1, 2, N_STEPS} if name == "fare_pgd" else {N_STEPS}
        for s, r in enumerate(rec.rows):
            if s + 1 in full:
                for k, v in r["_weights"].items():
                    arrs[f"{name}::w{s + 1}::{k}"] = v.numpy()
        arrs[f"{name}::wsum"] = np.array([[float(v.double().sum()) for v in r["_weights"].values()] for r in rec.rows])
        arrs[f"{name}::wsq"] = np.array([[float((v.double() ** 2).sum()) for v in r["_weights"].values()]
                                         for r in rec.rows])
        print(name, "lr used", lr_used, "loss", arrs[f"{name}::loss"], "eval", arrs[f"{name}::eval"])

    p = os.path.join(OUT, "train_step_tiny.npz")
    np.savez_compressed(p, **arrs)
    print(f"wrote train_step_tiny.npz: {os.path.getsize(p) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
