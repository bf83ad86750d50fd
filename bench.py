#!/usr/bin/env python3
"""bench.py - adversarial images/sec of the MI355X-native PGD inner loop (BASELINE.json metric).

One "step" = one complete ``pgd()`` call (10-step L-inf PGD, eps=4/255, step 1/255, FARE l2 loss,
delta0 ~ U(-eps,eps), mode='max') over one synthetic batch of 128 images 224x224x3 on the CLIP
ViT-L/14 vision encoder in bf16 (BASELINE config 2; config 4 = the same per GPU on 8 GPUs).
Weights are seeded random-init of that architecture (no checkpoints offline), inputs are resident in
HBM before the timed region.  One process per GPU; the attack is per-sample, so ranks shard images
with NO data-path collective (weak scaling) - torch.distributed (RCCL) only provides the barrier and
the max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - bf16 MFMA GEMM kernel: algorithmic FLOPs / HIP-event launch time, vs 2.5 PFLOP/s
  cpu_baseline - the oracle (PyTorch-CPU restatement of the reference path) timed on this host
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
FLOP_PER_IMG_ITER = {"ViT-L-14": 330.545e9, "ViT-B-32": 17.728e9}   # SURVEY.md Appendix C (fwd + input-bwd)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="ViT-L-14")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--attack", default="pgd", choices=["pgd", "apgd", "autopgd", "square"],
                    help="pgd: BASELINE configs 2/4 (FARE); apgd: config 3 (TeCoA apgd_train); autopgd: config 5 "
                         "(APGDAttack CE on the zero-shot head, use --iterations 100 --batch 256); square: the black-box "
                         "route (SquareAttack, --iterations = queries; forward passes only)")
    ap.add_argument("--mode", default="attack", choices=["attack", "train"],
                    help="attack (default, the BASELINE metric): one pgd()/apgd call per step; train: one full "
                         "FARE/TeCoA optimizer step per step (e0 + attack + fwd + wgrad backward + grad all-reduce + AdamW)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(iterations_full=10):
    """The reference path's CPU restatement (oracle/) on this host's cores: a bounded sample of the
    SAME workload (ViT-L/14 fp32, FARE PGD, eps=4/255), scaled to adversarial images/sec."""
    from oracle import vit_ref as V
    from oracle.attacks_ref import pgd_ref
    from oracle.losses_ref import ComputeLossWrapperRef
    # 32 threads: the small-M GEMMs of these batches do not scale past that (256 threads on the
    # GPU box's 2x64-core host was 50x SLOWER than 32: oversubscribed OpenMP teams on 514-row matmuls)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = V.VIT_L_14
    w = V.init_weights(cfg, seed=0)
    model = V.ClipVisionModelRef(cfg, w).eval()
    g = torch.Generator().manual_seed(0)
    eps = 4 / 255

    def sample(B, iters):
        x = torch.rand(B, 3, 224, 224, generator=g)
        d0 = torch.zeros_like(x).uniform_(-eps, eps, generator=g)
        with torch.no_grad():
            e0 = model(x, False)
        wrap = ComputeLossWrapperRef(e0, None, "mean", "l2", 100.)
        t0 = time.time()
        pgd_ref(model, wrap, x, None, "linf", eps, iters, 1 / 255, False, perturbation=d0, mode="max")
        return time.time() - t0

    sample(2, 1)                       # untimed: thread pool / allocator warm-up
    probe = sample(2, 1)               # sizes the timed sample to ~15 s of CPU work
    B = 16
    iters = int(max(1, min(iterations_full, round(15.0 / max(probe * B / 2, 1e-3)))))
    dt = sample(B, iters)
    per_call_full = dt * iterations_full / iters
    return {"value": B / per_call_full, "unit": "adversarial images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle pgd_ref (torch {torch.__version__} CPU fp32, {cores} threads): ViT-L/14, batch {B}, "
                      f"{iters} of {iterations_full} PGD iterations timed ({dt:.1f} s) after an untimed warm-up, scaled "
                      f"x{iterations_full / iters:.2g}; host has {os.cpu_count()} hardware threads"}


def bench_train(args, R, cfg, sd, dev, dist, world, rank):
    """Full training step (the 'next' row of SURVEY.md 8(f)): reported separately from the headline metric."""
    from robustvlm_amd.trainer import AdversarialTrainer
    B = args.batch
    tr = AdversarialTrainer(cfg, sd, batch_size=B, precision=args.precision, attack=args.attack if args.attack != "autopgd" else "pgd",
                            iterations_adv=args.iterations, device=dev)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        tr.train_step(x, None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.train_step(x, None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    barrier()
    if rank == 0:
        print(json.dumps({
            "metric": f"FARE training images/sec ({args.model}, {args.iterations}-step {args.attack} + optimizer step)",
            "value": world * B * args.steps / el, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"full train_one_epoch step: e0 + {args.attack} {args.iterations}-step + adv forward + "
                                   f"weight-gradient backward + flat-buffer all-reduce + AdamW, {args.model} {args.precision}, "
                                   f"batch={B} per GPU", "global_batch": world * B,
                       "parallelism": f"dp{world} (1 RCCL all-reduce of {tr.params.numel} fp32 grads per step)"},
            "final_loss": float(out["loss"])}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import robustvlm_amd as R
    cfg = R.CONFIGS[args.model]
    sd = R.random_state_dict(cfg, seed=0, device=dev)           # same weights on every rank
    B = args.batch
    if args.mode == "train":
        return bench_train(args, R, cfg, sd, dev, dist, world, rank)
    eng = R.VitEngine(cfg, sd, precision=args.precision, max_batch=args.batch, device=dev)
    del sd
    model = R.ClipVisionModel(eng).eval()
    g = torch.Generator(device=dev).manual_seed(1000 + rank)   # each rank its own shard of images
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g, device=dev)
    eps, stepsize = 4 / 255, 1 / 255
    d0 = (torch.rand(x.shape, generator=g, device=dev) * 2 - 1) * eps
    y = torch.randint(0, 1000, (B,), generator=g, device=dev)
    e0 = model(x, args.attack == "apgd")                        # embedding_orig (…clip.py:296-297)
    if args.attack == "pgd":
        wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)

        def step():
            return R.pgd(model, wrap, x, y, "linf", eps, args.iterations, stepsize, False,
                         perturbation=d0, mode="max")
    elif args.attack == "apgd":
        T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
        T = T / T.norm(dim=0, keepdim=True)
        wrap = R.ComputeLossWrapper(e0, T, "none", "ce", 100.)

        def step():
            return R.apgd_train(model, x, y, "linf", eps, n_iter=args.iterations, loss_fn=wrap)
    elif args.attack == "square":
        T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
        T = T / T.norm(dim=0, keepdim=True)
        clf = R.ClassificationModel(eng, T).eval()
        with torch.no_grad():
            y = clf(x).max(1)[1]
        atk = R.SquareAttack(clf, norm="Linf", n_queries=args.iterations, eps=eps, p_init=.8, n_restarts=1, seed=0,
                             resc_schedule=False)

        def step():
            return atk.perturb(x, y)
    else:
        T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
        T = T / T.norm(dim=0, keepdim=True)
        clf = R.ClassificationModel(eng, T).eval()
        with torch.no_grad():
            y = clf(x).max(1)[1]           # attack the clean predictions: every sample starts "correct"
        atk = R.APGDAttack(clf, n_iter=args.iterations, norm="Linf", n_restarts=1, eps=eps, seed=0, loss="ce",
                           alpha=2.0, use_rs=True)

        def step():
            return atk.perturb(x, y)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    barrier()
    assert float((out - x).abs().max()) <= 4 / 255 + 1e-6, "perturbation left the eps ball"

    res = None
    if rank == 0:
        value = world * B * args.steps / el
        res = {
            "metric": "adversarial images/sec (ViT-L/14, 10-step PGD eps=4/255)"
                      if (args.model == "ViT-L-14" and args.attack == "pgd" and args.iterations == 10)
                      else f"adversarial images/sec ({args.model}, {args.iterations}-step {args.attack})",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"{'FARE' if args.attack == 'pgd' else 'TeCoA-CE'} {args.attack.upper()} {args.iterations}-step eps=4/255 on {args.model} "
                                    f"{args.precision}, batch={B} per GPU, 224x224x3 synthetic, seeded random-init weights "
                                    f"(BASELINE configs[{(1 if world == 1 else 3) if args.attack == 'pgd' else 2 if args.attack == 'apgd' else 4}])")
                                   if args.attack != "square" else
                                   (f"black-box SquareAttack, {args.iterations} queries, eps=4/255 on {args.model} {args.precision} + "
                                    f"1000-class zero-shot head, batch={B} per GPU (clip_robustbench.py --blackbox_only route; "
                                    f"forward passes only, samples leave the batch once fooled)"),
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world} (no data-path collective)",
                       "loss": "l2/mean" if args.attack == "pgd" else "margin" if args.attack == "square" else "ce/none"},
            # pgd: I x (fwd+bwd); apgd_train: (I+1) fwd + I bwd; APGDAttack: (I+2) fwd + (I+1) bwd  ~ I+1 pairs;
            # square: at most I + 3 forwards (0.49 of a pair each; fewer once samples are fooled)
            "whole_loop_tflops_per_gpu": value / world * FLOP_PER_IMG_ITER.get(args.model, 0) *
                                         (args.iterations if args.attack == "pgd" else args.iterations + 0.5 if
                                          args.attack == "apgd" else 0.49 * (args.iterations + 3) if
                                          args.attack == "square" else args.iterations + 1.5) / 1e12,
        }
        res["whole_loop_frac_of_peak"] = res["whole_loop_tflops_per_gpu"] / PEAK_BF16_TFLOPS
        res["whole_loop_flop_basis"] = ("reference model FLOPs per image (SURVEY.md Appendix C); the engine's class-token "
                                        "tail skips the dead rows of the last block (~3 % of them) - roofline.achieved "
                                        "counts executed FLOPs only")

    # ---- roofline of the dominant kernel (bf16 MFMA GEMM), HIP events on the engine's stream ------
    if rank == 0 and not args.no_roofline:
        eng.set_profiling(True)
        eng.reset_profile()
        step()
        prof = eng.get_profile()
        eng.set_profiling(False)
        gemm = {k: v for k, v in prof.items() if k.startswith("gemm_") and "patch" not in k}
        gflops = sum(v["flops"] for v in gemm.values())
        gms = sum(v["ms"] for v in gemm.values())
        glaunch = sum(v["launches"] for v in gemm.values())
        attn_gemm = {k: v for k, v in prof.items() if k in ("gemm_qkv_fwd", "gemm_out_fwd", "gemm_qkv_bwd",
                                                            "gemm_out_bwd", "attn_fwd", "attn_bwd")}
        achieved = gflops / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
        # HBM-side traffic comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be
        # read from inside this process); the committed measurement of this same command is reported.
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if os.path.exists(tj) and args.model == "ViT-L-14" and args.batch == 128 and args.precision == "bf16":
            try:
                traffic = json.load(open(tj))["gemm_bytes_per_logical_launch"]
                traffic_src = "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
            except Exception:
                traffic = None
        res["roofline"] = {
            "bound": "mfma", "kernel": "gemm_bf16_nt_256p_kernel (QKV/out-proj/fc1/fc2, fwd + dgrad; the 128 remainder rows ride in the same launch)",
            "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes per GEMM launch (fabric side: 2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "flops_per_launch": gflops / max(glaunch, 1), "avg_launch_ms": gms / max(glaunch, 1),
            "launches": glaunch,
            "attention_gemm_subset": {
                "tflops": sum(v["flops"] for v in attn_gemm.values()) / max(sum(v["ms"] for v in attn_gemm.values()), 1e-9) / 1e9,
                "frac": sum(v["flops"] for v in attn_gemm.values()) / max(sum(v["ms"] for v in attn_gemm.values()), 1e-9) / 1e9 / PEAK_BF16_TFLOPS},
            "per_class": {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                              "gbps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] > 0 else None}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.iterations)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
