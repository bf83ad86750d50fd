#!/usr/bin/env python3
"""bench.py - adversarial images/sec of the MI355X-native PGD inner loop (BASELINE.json metric).

One "step" = one complete ``pgd()`` call (10-step L-inf PGD, eps=4/255, step 1/255, FARE l2 loss,
delta0 ~ U(-eps,eps), mode='max') over one synthetic batch of 128 images 224x224x3 on the CLIP
ViT-L/14 vision encoder in bf16 (BASELINE config 2; config 4 = the same per GPU on 8 GPUs).
Weights are seeded random-init of that architecture (no checkpoints offline), inputs are resident in
HBM before the timed region.  One process per GPU; the attack is per-sample, so ranks shard images
with NO data-path collective (weak scaling) - torch.distributed (RCCL) only provides the barrier and
the max-over-ranks time.

`python bench.py --gpus N` started WITHOUT a launcher creates its N ranks itself (robustvlm_amd/launch.py: re-exec under
torch.distributed.run on 127.0.0.1, exit code of the ranks); started under one (WORLD_SIZE set) it is a rank, and --gpus has
to equal WORLD_SIZE.  More GPUs than the node shows is an error, never a silently smaller job.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - bf16 MFMA GEMM kernel: algorithmic FLOPs / HIP-event launch time, vs 2.5 PFLOP/s; the shader clock held
                 during the timed region (rocm-smi) and inside the GEMM launches (the kernel's own cycle / real-time stamps),
                 the fraction against the peak at that clock, per-class times, the committed in-pipeline PMC summary
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
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from benchlib import ClockSampler, bind_rank_to_numa, hbm_classes, host_cpu_info, pmc_in_run  # noqa: E402  (untimed measurement helpers)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3   # fp32-input MFMA (= the fp32 vector peak), same guide: --precision fp32 runs are priced against it
FLOP_PER_IMG_ITER = {"ViT-L-14": 330.545e9, "ViT-B-32": 17.728e9}   # SURVEY.md Appendix C (fwd + input-bwd)


HEADLINE_MODELS = ("ViT-L-14", "ViT-B-32")      # the models BASELINE.json's configs name (B/32: configs[0] through the HIP path)


def parse(extra=None):
    """Options of the headline harness: BASELINE configs 2/4 (pgd), 3 (apgd), 5 (autopgd) and the training step.  Everything no
    BASELINE config asks for (the L2 threat model, SquareAttack, other models) is added by scripts/bench_extra.py through
    `extra` - this file alone cannot run them."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="ViT-L-14", choices=list(HEADLINE_MODELS) + list(getattr(extra, "MODELS", ())))
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "x3", "bf16+fp32-first", "bf16+x3-first", "bf16+x3fwd-first", "bf16+x3fwd"],
                    help="bf16: the throughput path (headline); fp32: the reference's own precision on the fp32 matrix-pipe tiles; "
                         "x3: fp32 storage, the linears as split-bf16 products (3 bf16 MFMAs per product, ~16 mantissa bits); "
                         "bf16+fp32-first / bf16+x3-first: pgd with its first iteration (and the clean embedding) on that handle")
    ap.add_argument("--attack", default="pgd", choices=["pgd", "apgd", "autopgd"] + list(getattr(extra, "ATTACKS", ())),
                    help="pgd: BASELINE configs 2/4 (FARE); apgd: config 3 (TeCoA apgd_train); autopgd: config 5 "
                         "(APGDAttack CE on the zero-shot head, use --iterations 100 --batch 256)")
    ap.add_argument("--mode", default="attack", choices=["attack", "train"],
                    help="attack (default, the BASELINE metric): one pgd()/apgd call per step; train: one full "
                         "FARE/TeCoA optimizer step per step (e0 + attack + fwd + wgrad backward + grad all-reduce + AdamW)")
    ap.add_argument("--always-reduce", action="store_true",
                    help="train mode on ONE GPU: run the bucketed RCCL all-reduce path in a one-rank group (identity reduction) "
                         "so that its launch / wait structure and the `allreduce` diagnostics can be exercised without a node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the in-run rocprofv3 --pmc passes (fabric traffic, matrix-pipe busy): roofline.traffic / "
                         "pmc_in_pipeline then come from the committed profiles/ files, flagged measured_in_run: false")
    if extra is not None:
        extra.add_arguments(ap)
    args = ap.parse_args()
    if not hasattr(args, "norm"):
        args.norm = "linf"
    return args


def cpu_baseline(iterations_full=10, l14_sample=True):
    """The reference path's CPU restatement (oracle/, parity-pinned against the imported reference) on this host's
    cores, as BASELINE.md section 4 specifies: config 1 = FARE PGD 10-step eps=4/255 on ViT-B/32, batch 8, fp32,
    torch.rand images (seed 0), delta0 ~ U(-eps, eps) (seed 1); 3 warm-ups, median of 10 full pgd_ref calls.
    Thread count: the fastest of {physical cores, 32, 16} in a one-call probe (the 400-row matmuls of this batch do
    not scale to 128 threads); `cores` is the count actually used.  A bounded ViT-L/14 sample (the GPU workload's
    model) is reported next to it."""
    from oracle import vit_ref as V
    from oracle.attacks_ref import pgd_ref
    from oracle.losses_ref import ComputeLossWrapperRef
    cpu_model, physical, threads = host_cpu_info()
    eps = 4 / 255

    def problem(cfg, B):
        w = V.init_weights(cfg, seed=0)
        model = V.ClipVisionModelRef(cfg, w).eval()
        x = torch.rand(B, 3, 224, 224, generator=torch.Generator().manual_seed(0))
        d0 = torch.zeros_like(x).uniform_(-eps, eps, generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            e0 = model(x, False)
        return model, ComputeLossWrapperRef(e0, None, "mean", "l2", 100.), x, d0

    def call(pb, iters):
        model, wrap, x, d0 = pb
        t0 = time.time()
        pgd_ref(model, wrap, x, None, "linf", eps, iters, 1 / 255, False, perturbation=d0.clone(), mode="max")
        return time.time() - t0

    # ---- config 1 ----
    B1 = 8
    pb = problem(V.VIT_B_32, B1)
    probe = {}
    for n in sorted({physical, 32, 16}, reverse=True):
        if n > threads:
            continue
        torch.set_num_threads(n)
        call(pb, 1)
        probe[n] = call(pb, 2)
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    for _ in range(3):
        call(pb, iterations_full)
    N_CALLS = 10                                     # BASELINE.md section 4 / SURVEY.md 8(d): median of >= 10
    times = sorted(call(pb, iterations_full) for _ in range(N_CALLS))
    med = 0.5 * (times[(N_CALLS - 1) // 2] + times[N_CALLS // 2])
    out = {"value": B1 / med, "unit": "adversarial images/sec", "cores": cores, "kind": "port",
           "sample": f"BASELINE config 1: oracle pgd_ref (torch {torch.__version__} CPU fp32), FARE PGD {iterations_full}-step "
                     f"eps=4/255, ViT-B/32, batch {B1}; 3 warm-ups, median of {N_CALLS} full calls ({med:.2f} s, min {times[0]:.2f}, "
                     f"max {times[-1]:.2f}); {cores} threads = fastest of a probe over "
                     f"{ {k: round(v, 2) for k, v in probe.items()} } (s per 2 iterations)",
           "cpu_model": cpu_model, "physical_cores": physical, "hardware_threads": threads}
    # ---- the GPU workload's own model, bounded sample ----
    if l14_sample:
        torch.set_num_threads(min(threads, 32))
        pb = problem(V.VIT_L_14, 16)
        t1 = call(pb, 1)                       # warm-up iteration (allocator, thread pool)
        iters = int(max(1, min(iterations_full, round(12.0 / max(t1, 1e-3)))))
        dt = call(pb, iters)
        out["vit_l14_sample"] = {"value": 16 / (dt * iterations_full / iters), "unit": "adversarial images/sec",
                                 "cores": min(threads, 32),
                                 "sample": f"ViT-L/14, batch 16, {iters} of {iterations_full} PGD iterations timed ({dt:.1f} s) "
                                           f"after one warm-up iteration, scaled x{iterations_full / iters:.2g}"}
    return out


REHEARSAL_NOTE = ("RVLM_BENCH_REHEARSAL=gloo: the ranks of this job share GPU(s) and meet over gloo - the line proves the "
                  "world > 1 code path (self-launch, process group, barriers, gather of per-rank times), it is NOT a measurement")
def _baseline_config_name(attack: str, world: int, per_gpu_batch: int, model: str = "ViT-L-14") -> str:
    """Which BASELINE.json `configs` entry a run is: [1] one GPU, [3] = the same 128 images per GPU on 8 GPUs
    (global batch 1024); 2 / 4 GPUs are the intermediate points of the 1/2/4/8 curve of that same per-GPU workload."""
    if model == "ViT-B-32":
        return ("BASELINE configs[0]'s problem through the HIP path" if attack == "pgd" and per_gpu_batch == 8 and world == 1
                else "ViT-B/32: the model of BASELINE configs[0], not its batch / attack")
    if model != "ViT-L-14":
        return "no BASELINE config names this model"
    if attack == "apgd":
        return "BASELINE configs[2]" + ("" if world == 1 else f" per GPU x {world}")
    if attack != "pgd":
        return "BASELINE configs[4], n_restarts=1 (the reference's AutoAttack default is 5)" + ("" if world == 1 else f" per GPU x {world}")
    if world == 1:
        return "BASELINE configs[1]"
    if world == 8 and per_gpu_batch == 128:
        return "BASELINE configs[3]: global batch 1024 = 8 x 128, data-parallel"
    return f"BASELINE configs[1] per GPU x {world} ranks: a point of the 1/2/4/8 curve that ends at configs[3]"


def bench_train(args, R, cfg, sd, dev, dist, world, rank, coll_dev=None, rehearsal=""):
    """Full training step (the 'next' row of SURVEY.md 8(f)): reported separately from the headline metric."""
    from robustvlm_amd.trainer import AdversarialTrainer
    B = args.batch
    tr = AdversarialTrainer(cfg, sd, batch_size=B, precision=args.precision, attack=args.attack if args.attack != "autopgd" else "pgd",
                            iterations_adv=args.iterations, device=dev, always_reduce=args.always_reduce)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        tr.train_step(x, None, global_batch=world * B)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.train_step(x, None, global_batch=world * B)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=coll_dev if coll_dev is not None else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    barrier()
    # with more than one rank (or --always-reduce on one): per-bucket all-reduce times and how much of them the backward hid
    ar = tr.profile_allreduce(x) if tr._reduce else None
    if rank == 0:
        print(json.dumps({
            "allreduce": ar, **({"rehearsal": REHEARSAL_NOTE} if rehearsal else {}),
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


def in_kernel_clock(step, dev):
    """Effective shader clock INSIDE the GEMM launches of one more step: the persistent kernel's trace hook makes wave 0 of every
    workgroup stamp s_memtime (shader clock) and s_memrealtime (constant 100 MHz) at its start and end."""
    try:
        from robustvlm_amd import _lib as L
        tr = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
        L.load().rvlm_k_gemm_set_trace(tr.data_ptr())
        step()
        torch.cuda.synchronize()
        L.load().rvlm_k_gemm_set_trace(None)
        t = tr.view(256, 8, 4)[:, 7, :].double()          # [wg][memtime0, realtime0, memtime1, realtime1] of the LAST launch
        ok = (t[:, 3] > t[:, 1]) & (t[:, 2] > t[:, 0])
        if bool(ok.any()):
            mhz = ((t[ok, 2] - t[ok, 0]) / (t[ok, 3] - t[ok, 1]) * 100.0)
            return {"sclk_mhz_effective": float(mhz.mean()), "workgroups": int(ok.sum()),
                    "source": "s_memtime / s_memrealtime stamps of the last persistent-GEMM launch of a pgd() call "
                              "(rvlm_k_gemm_set_trace)"}
        return None
    except Exception as e:                     # measurement aid only
        return {"error": f"{type(e).__name__}: {e}"}


def main(extra=None):
    args = parse(extra)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # `python bench.py --gpus N` without a launcher creates its N ranks itself (and exits with their exit code);
    # under a launcher (WORLD_SIZE set) this process is one rank and --gpus must agree with it
    from robustvlm_amd.launch import ensure_ranks, rehearsal_backend
    plan = ensure_ranks(args.gpus)
    world = plan.world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RVLM_BENCH_REHEARSAL=gloo (robustvlm_amd/launch.py): the world > 1 branch on a node with fewer GPUs than ranks - rank r
    # on device r mod n, gloo instead of RCCL.  Exercises every line below except the RCCL-only ones;
    # its JSON line says so and is not a measurement.
    rehearsal = rehearsal_backend(os.environ) if world > 1 else ""
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    affinity = bind_rank_to_numa(dev_index, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    dist = None
    coll_dev = dev                     # device of the tensors handed to the metric collectives (gloo: host)
    if world > 1 or (args.mode == "train" and args.always_reduce):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:                      # one-rank group (--always-reduce): any free local port
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            coll_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import robustvlm_amd as R
    cfg = R.CONFIGS[args.model]
    sd = R.random_state_dict(cfg, seed=0, device=dev)           # same weights on every rank
    B = args.batch
    if args.mode == "train":
        return bench_train(args, R, cfg, sd, dev, dist, world, rank, coll_dev, rehearsal)
    eng = R.VitEngine(cfg, sd, precision=args.precision, max_batch=args.batch, device=dev)
    del sd
    model = R.ClipVisionModel(eng).eval()
    # x = torch.rand seed 0, delta0 ~ U(-eps, eps) seed 1, y = randint seed 2 - drawn by the DEVICE generator
    # (the parity tests draw the same seeds on the CPU generator: other streams, the same distribution), + 1000 * rank so
    # that every rank holds its own shard of images
    seeds = {"x": 0 + 1000 * rank, "delta0": 1 + 1000 * rank, "y": 2 + 1000 * rank}
    g = torch.Generator(device=dev).manual_seed(seeds["x"])
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g, device=dev)
    eps, stepsize, eps_txt = 4 / 255, 1 / 255, "4/255"
    d0 = (torch.rand(x.shape, generator=torch.Generator(device=dev).manual_seed(seeds["delta0"]), device=dev) * 2 - 1) * (4 / 255)
    y = torch.randint(0, 1000, (B,), generator=torch.Generator(device=dev).manual_seed(seeds["y"]), device=dev)
    e0 = model(x, args.attack == "apgd")                        # embedding_orig (…clip.py:296-297)
    # pairs of (forward + input backward) one step runs, for whole_loop_tflops: pgd I; apgd_train (I+1) fwd + I bwd;
    # APGDAttack (I+2) fwd + (I+1) bwd ~ I + 1.5
    pairs = {"pgd": args.iterations, "apgd": args.iterations + 0.5, "autopgd": args.iterations + 1.5}.get(args.attack)
    workload, loss_name = None, "l2/mean" if args.attack == "pgd" else "ce/none"
    in_ball = lambda out: float((out - x).abs().max()) <= 4 / 255 + 1e-6        # noqa: E731
    if extra is not None and extra.handles(args):
        # a workload no BASELINE config names (scripts/bench_extra.py): same harness, its own step / labels
        step, workload, loss_name, pairs, in_ball = extra.make_step(args, R, eng, model, cfg, x, y, d0, e0, g, dev)
    elif args.attack == "pgd":
        wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)

        def step():
            return R.pgd(model, wrap, x, y, "linf", eps, args.iterations, stepsize, False, perturbation=d0, mode="max")
    elif args.attack == "apgd":
        T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
        T = T / T.norm(dim=0, keepdim=True)
        wrap = R.ComputeLossWrapper(e0, T, "none", "ce", 100.)

        def step():
            return R.apgd_train(model, x, y, "linf", eps, n_iter=args.iterations, loss_fn=wrap)
    else:
        T = torch.randn(cfg.out_dim, 1000, generator=g, device=dev)
        T = T / T.norm(dim=0, keepdim=True)
        clf = R.ClassificationModel(eng, T).eval()
        with torch.no_grad():
            y = clf(x).max(1)[1]           # attack the clean predictions: every sample starts "correct"
        atk = R.APGDAttack(clf, n_iter=args.iterations, norm="Linf", n_restarts=1, eps=eps, seed=0, loss="ce", alpha=2.0, use_rs=True)

        def step():
            return atk.perturb(x, y)
    if workload is None:
        workload = (f"{'FARE' if args.attack == 'pgd' else 'TeCoA-CE'} {args.attack.upper()} {args.iterations}-step eps={eps_txt} on "
                    f"{args.model} {args.precision}, batch={B} per GPU, 224x224x3 synthetic, seeded random-init weights "
                    f"({_baseline_config_name(args.attack, world, B, args.model)})")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    if os.environ.get("RVLM_BENCH_DIE_RANK") == str(rank):      # test hook: a rank that dies mid-job (tests/test_gpu_bench_entry.py)
        os._exit(17)
    barrier()
    # the timed region: exactly --steps calls between two barrier + synchronize pairs, nothing else running in this process
    # (the rocm-smi clock sampler of round 3 now polls during extra UNTIMED steps below).  Device-side events between the
    # calls give the per-call distribution (median of the K calls) without a host sync.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # Every rank then looks at its OWN GPU (untimed extra calls, all ranks at once so that the node is loaded as in the timed
    # region): rocm-smi clock / socket power of its device and the shader clock its GEMM launches actually got.  The boxes of this
    # pool hold 1.74-1.92 GHz under the power cap; a 1 -> 8 curve has to show whether a slow rank is a slow GPU.
    sampler, my_clock = None, None
    if not args.no_roofline:
        with ClockSampler(dev_index) as sampler:
            for _ in range(min(args.steps, 3)):
                step()
            torch.cuda.synchronize()
        my_clock = in_kernel_clock(step, dev)
    smi = sampler.summary() if sampler is not None else None
    mine = {"rank": rank, "device": dev_index, "seconds": el, "images_per_sec": B * args.steps / el,
            "sclk_in_kernel_mhz": (my_clock or {}).get("sclk_mhz_effective"),
            "sclk_smi_mhz_mean": (smi or {}).get("sclk_mhz_mean"), "socket_power_w_mean": (smi or {}).get("socket_power_w_mean"),
            "cpu_affinity": affinity}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        per_rank_s = [r["seconds"] for r in per_rank]
        el = max(per_rank_s)                                 # the job is as slow as its slowest rank
    else:
        per_rank_s = [el]
    barrier()
    assert in_ball(out), "perturbation left the eps ball"

    res = None
    if rank == 0:
        value = world * B * args.steps / el
        res = {
            "metric": "adversarial images/sec (ViT-L/14, 10-step PGD eps=4/255)"
                      if (args.model == "ViT-L-14" and args.attack == "pgd" and args.iterations == 10 and args.norm == "linf")
                      else f"adversarial images/sec ({args.model}, {args.iterations}-step {args.attack}{'' if args.norm == 'linf' else ', L2'})",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            # `value` = all K calls / the bracketed wall time (the contract's definition = mean); the median call of the
            # same K, from device events on this rank's stream, next to it (median of >= 10 timed calls)
            "ms_per_step_median": 0.5 * (step_ms[(len(step_ms) - 1) // 2] + step_ms[len(step_ms) // 2]),
            "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
            "value_median_call": B * 1e3 / (0.5 * (step_ms[(len(step_ms) - 1) // 2] + step_ms[len(step_ms) // 2])) * world,
            "config": {"workload": workload,
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world} (no data-path collective)",
                       "loss": loss_name, "norm": args.norm,
                       "input_seeds": {**seeds, "generator": "torch device generator, rank 0 (rank r: + 1000 r)"}},
            "whole_loop_tflops_per_gpu": value / world * FLOP_PER_IMG_ITER.get(args.model, 0) * pairs / 1e12,
        }
        if rehearsal:
            res["rehearsal"] = REHEARSAL_NOTE
            res["data"] = "synthetic (REHEARSAL: ranks share a GPU over gloo - not a measurement)"
        res["ranks"] = {"rccl_ranks": world if (dist is not None and not rehearsal) else 0,
                        "gloo_ranks": world if rehearsal else 0,
                        "launcher": "none (single process)" if world == 1 else
                                    "bench.py self-launch (torch.distributed.run, 127.0.0.1)" if os.environ.get("RVLM_SELF_LAUNCHED")
                                    else "external torch.distributed.run",
                        "per_rank_images_per_sec_min": min(B * args.steps / t for t in per_rank_s),
                        "per_rank_images_per_sec_max": max(B * args.steps / t for t in per_rank_s),
                        # `value` is the contract's number: all images / the SLOWEST rank's time.  The sum of the ranks' own rates
                        # is what the GPUs delivered; the gap between the two is rank imbalance (clock / power per rank below),
                        # not communication - the attack has no data-path collective
                        "images_per_sec_sum_of_rank_rates": sum(B * args.steps / t for t in per_rank_s),
                        "cpu_affinity_rank0": affinity,
                        "per_rank": per_rank}
        res["whole_loop_frac_of_peak"] = res["whole_loop_tflops_per_gpu"] / (PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS)
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
        eff_clock = my_clock
        gemm = {k: v for k, v in prof.items() if k.startswith("gemm_") and "patch" not in k}
        gflops = sum(v["flops"] for v in gemm.values())
        gms = sum(v["ms"] for v in gemm.values())
        glaunch = sum(v["launches"] for v in gemm.values())
        attn_gemm = {k: v for k, v in prof.items() if k in ("gemm_qkv_fwd", "gemm_out_fwd", "gemm_qkv_bwd",
                                                            "gemm_out_bwd", "attn_fwd", "attn_bwd")}
        achieved = gflops / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
        # HBM-side traffic comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be
        # read from inside this process); the committed measurement of this same command is reported.
        traffic, traffic_src, pmc = None, None, None
        headline = (args.model == "ViT-L-14" and args.batch == 128 and args.precision == "bf16" and args.attack == "pgd"
                    and args.norm == "linf")
        pmc_error = None
        if headline and world == 1 and not args.no_pmc and not os.environ.get("RVLM_BENCH_CHILD"):
            try:
                torch.cuda.synchronize()
                traffic, traffic_src, pmc = pmc_in_run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--no-pmc",
                                                        "--iterations", str(args.iterations)])
            except Exception as e:                    # measurement aid: fall back to the committed passes, flagged as such
                pmc_error = f"{type(e).__name__}: {e}"[:400]
                traffic, traffic_src, pmc = None, None, None
        for tag in ("r06", "r05", "r04", "r03", "r02"):
            tj = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json")
            if traffic is None and os.path.exists(tj) and headline:
                try:
                    tjd = json.load(open(tj))
                    traffic = tjd["gemm_bytes_per_logical_launch"]
                    traffic_src = {"measured_in_run": False, "file": f"profiles/{tag}_pmc_hbm_traffic.json",
                                   "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command "
                                          "(scripts/pmc_traffic.sh) on another box of the pool, committed with the round's "
                                          "profiles; PMC counters cannot be read from inside this process",
                                   "collected_at_commit": tjd.get("collected_at_commit", f"round {tag[1:]} tree"),
                                   "dram_vs_infinity_cache": tjd.get("dram_vs_infinity_cache")}
                except Exception:
                    traffic = None
        # matrix-pipe utilisation INSIDE this pipeline (not of a cube): rocprofv3 --pmc passes over this same command,
        # summarised per kernel by scripts/pmc_pipeline.sh (counters cannot be read from inside this process either)
        pj = next((q for q in (os.path.join(ROOT, "profiles", f"{t}_pmc_pipeline.json") for t in ("r06", "r05", "r04", "r03")) if os.path.exists(q)), "")
        if pj and headline and pmc is None:
            try:
                kd = json.load(open(pj))["kernels"]
                pmc = {"measured_in_run": False,
                       "source": f"profiles/{os.path.basename(pj)} (scripts/pmc_pipeline.sh: rocprofv3 --pmc passes over bench.py --steps 1, "
                                 "another box of the pool, committed with that round's profiles)",
                       "mfma_busy": {k: v.get("mfma_busy") for k, v in kd.items() if v.get("mfma_busy") is not None},
                       "lds_conflict_share": {k: v.get("lds_conflict_share") for k, v in kd.items()
                                              if v.get("lds_conflict_share") is not None}}
            except Exception:
                pmc = None
        fp32_run = args.precision == "fp32"
        peak = PEAK_F32_MFMA_TFLOPS if fp32_run else PEAK_BF16_TFLOPS
        res["roofline"] = {
            "bound": "mfma",
            "kernel": "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32 tiles: QKV/out-proj/fc1/fc2, fwd + dgrad)" if fp32_run else
                      "gemm_bf16_nt_256p_kernel (QKV/out-proj/fc1/fc2, fwd + dgrad; the 128 remainder rows ride in the same launch)"
                      + (" - the bf16 handle's launches only (iterations 2..I); the first iteration's handle is not in this object"
                         if "+" in args.precision else "")
                      + (" - split-bf16 linears: `achieved` counts MODEL FLOPs (2 M N K per linear, split + activation passes in its "
                         "time); the kernel executes 3x that in bf16 MFMAs" if args.precision == "x3" else ""),
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_unit": "bytes per GEMM launch (fabric side: 2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": traffic_src, "pmc_in_run_error": pmc_error,
            "flops_per_launch": gflops / max(glaunch, 1), "avg_launch_ms": gms / max(glaunch, 1),
            "launches": glaunch,
            "pmc_in_pipeline": pmc,
            "clock_in_kernel": eff_clock,
            "attention_gemm_subset": {
                "tflops": sum(v["flops"] for v in attn_gemm.values()) / max(sum(v["ms"] for v in attn_gemm.values()), 1e-9) / 1e9,
                "frac": sum(v["flops"] for v in attn_gemm.values()) / max(sum(v["ms"] for v in attn_gemm.values()), 1e-9) / 1e9 / peak},
            # the HBM-bound quarter of the step against ITS roofline: algorithmic bytes / HIP-event time / 8 TB/s
            # (attention: q, k, v in + o out forward; q, k, v, o, dO in + dq, dk, dv out backward, bf16; LayerNorm: the
            # byte counts the engine's profile scopes carry - 6 B per element forward, 16 B backward)
            "hbm_classes": hbm_classes(prof, cfg, B, (pmc or {}).get("hbm_class_traffic")),
            "per_class_source": "ONE separate profiled pgd() call after the timed region, HIP events around every scope of the "
                                "engine: the attack-update kernels and the host-side gaps between scopes are in no class, the events "
                                "themselves cost the call ~1 %, so the class sum lands within ~1 % of ms_per_step on either side",
            "per_class": {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                              "gbps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] > 0 else None}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }
    if rank == 0 and sampler is not None and "roofline" in res:
        clk = smi
        res["roofline"]["clock"] = clk
        eff = (res["roofline"].get("clock_in_kernel") or {}).get("sclk_mhz_effective")
        if clk or eff:
            # the dense peak = 256 CUs x 2.4 GHz x (bf16 FLOP per CU and cycle): it scales linearly with the shader clock;
            # the in-kernel measurement (cycles the waves actually got) is preferred over rocm-smi's reading
            peak_at_clock = res["roofline"]["peak"] * (eff if eff else clk["sclk_mhz_mean"]) / 2400.0
            res["roofline"]["frac_at_measured_clock"] = res["roofline"]["achieved"] / peak_at_clock
            res["roofline"]["peak_at_measured_clock"] = peak_at_clock
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.iterations)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
