#!/usr/bin/env python3
"""bench_extra.py - bench.py's harness on workloads NO BASELINE config names (moved out of bench.py in round 6, VERDICT r5
item 8, so that the headline harness only holds what BASELINE.json asks for):

  --norm l2               the L2 threat model of pgd / apgd_train / APGDAttack (eps = 3.0, PGD step eps/4)
  --attack square         the black-box route (SquareAttack L-inf; forward passes only)
  --model ViT-L-14-336 .. every other entry of robustvlm_amd.CONFIGS

Same timed region, same JSON line (bench.main); `config.workload` says which of these it is and that it is not a BASELINE
config.  Usage: python scripts/bench_extra.py --norm l2 --attack apgd ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Extra:
    ATTACKS = ("square",)

    @property
    def MODELS(self):
        import robustvlm_amd as R
        return tuple(k for k in R.CONFIGS if k not in bench.HEADLINE_MODELS)

    def add_arguments(self, ap):
        ap.add_argument("--norm", default="linf", choices=["linf", "l2"],
                        help="threat model of pgd / apgd / autopgd (the --norm of the reference's trainer and of "
                             "CLIP_eval/clip_robustbench.py); linf: eps = 4/255; l2: eps = 3.0, PGD step eps/4")

    def handles(self, args):
        if args.norm != "linf" and args.attack == "square":
            raise SystemExit("--attack square is L-inf only")
        return args.norm != "linf" or args.attack == "square"

    def make_step(self, args, R, eng, model, cfg, x, y, d0, e0, g, dev):
        """-> (step, config.workload, config.loss, (forward + input backward) pairs per step, in_ball(out))"""
        B, it = x.shape[0], args.iterations
        if args.attack == "square":
            T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
            T = T / T.norm(dim=0, keepdim=True)
            clf = R.ClassificationModel(eng, T).eval()
            with torch.no_grad():
                yc = clf(x).max(1)[1]
            atk = R.SquareAttack(clf, norm="Linf", n_queries=it, eps=4 / 255, p_init=.8, n_restarts=1, seed=0, resc_schedule=False)
            text = (f"black-box SquareAttack, {it} queries, eps=4/255 on {args.model} {args.precision} + 1000-class zero-shot head, "
                    f"batch={B} per GPU (clip_robustbench.py --blackbox_only route; forward passes only, samples leave the batch "
                    f"once fooled); no BASELINE config")
            # at most it + 3 forwards (0.49 of a pair each; fewer once samples are fooled)
            return (lambda: atk.perturb(x, yc)), text, "margin", 0.49 * (it + 3), lambda out: float((out - x).abs().max()) <= 4 / 255 + 1e-6
        eps, stepsize = 3.0, 0.75
        # the start point INSIDE the L2 ball (U(-4/255, 4/255) per pixel has |d0|_2 ~ 3.5 > eps, so the first
        # iteration - the one 'bf16+fp32-first' is built around - was evaluated outside the threat model)
        d0 = R.project_perturbation(d0, eps, "l2")
        in_ball = lambda out: float((out - x).flatten(1).norm(dim=1).max()) <= eps * (1 + 1e-5)      # noqa: E731
        tail = "the L2 threat model of the same entry points; no BASELINE config"
        if args.attack == "pgd":
            wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)
            step = lambda: R.pgd(model, wrap, x, y, "l2", eps, it, stepsize, False, perturbation=d0, mode="max")   # noqa: E731
            pairs, loss = it, "l2/mean"
        elif args.attack == "apgd":
            T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
            T = T / T.norm(dim=0, keepdim=True)
            wrap = R.ComputeLossWrapper(e0, T, "none", "ce", 100.)
            step = lambda: R.apgd_train(model, x, y, "l2", eps, n_iter=it, loss_fn=wrap)                          # noqa: E731
            pairs, loss = it + 0.5, "ce/none"
        else:
            T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
            T = T / T.norm(dim=0, keepdim=True)
            clf = R.ClassificationModel(eng, T).eval()
            with torch.no_grad():
                yc = clf(x).max(1)[1]
            atk = R.APGDAttack(clf, n_iter=it, norm="L2", n_restarts=1, eps=eps, seed=0, loss="ce", alpha=2.0, use_rs=True)
            step = lambda: atk.perturb(x, yc)                                                                    # noqa: E731
            pairs, loss = it + 1.5, "ce/none"
        text = (f"{'FARE' if args.attack == 'pgd' else 'TeCoA-CE'} {args.attack.upper()} {it}-step eps=3.0 (L2) on {args.model} "
                f"{args.precision}, batch={B} per GPU, 224x224x3 synthetic, seeded random-init weights ({tail})")
        return step, text, loss, pairs, in_ball


if __name__ == "__main__":
    bench.main(Extra())
