"""bench.py as the driver starts it: one command, N GPUs - or a loud failure."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


BENCH_EXTRA = os.path.join(ROOT, "scripts", "bench_extra.py")     # the workloads no BASELINE config names (L2, Square, other models)


def run(*a, env=None, timeout=600, script=BENCH):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, script, *a], capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


def test_more_gpus_than_devices_exits_nonzero():
    n = torch.cuda.device_count() + 1
    r = run("--gpus", str(n), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert f"--gpus {n} but only {n - 1} GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no throughput line may be printed"


def test_world_size_mismatch_exits_nonzero():
    r = run("--gpus", "8", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr


def test_single_gpu_line(tmp_path):
    """N = 1 prints ONE JSON line with the contract's keys (small model so that the test stays short)."""
    r = run("--gpus", "1", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["ranks"]["rccl_ranks"] == 0 and d["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_gpu_self_launch():
    r = run("--gpus", "2", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["ranks"]["rccl_ranks"] == 2 and "self-launch" in d["ranks"]["launcher"]


# ---- the world > 1 branch of bench.py BEFORE an 8-GPU node runs it -----------------------------------
# RVLM_BENCH_REHEARSAL=gloo: two ranks share cuda:0 and meet over gloo.  Everything of the N > 1 path runs except the
# RCCL-only lines (init_process_group("nccl", device_id=...), device-tensor collectives): self-launch under
# torch.distributed.run on 127.0.0.1, per-rank engines and image shards, barriers, the gather of per-rank times, rank 0's
# ONE JSON line, the agent's exit code.  Reference's multi-GPU: train/adversarial_training_clip.py:184-191.
REH = {"RVLM_BENCH_REHEARSAL": "gloo"}
SMALL = ("--steps", "2", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--no-cpu-baseline")


def _one_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_attack_line():
    d = _one_line(run("--gpus", "2", *SMALL, env=REH))
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "rehearsal" in d and "REHEARSAL" in d["data"]
    rk = d["ranks"]
    assert rk["gloo_ranks"] == 2 and rk["rccl_ranks"] == 0 and "self-launch" in rk["launcher"]
    assert 0 < rk["per_rank_images_per_sec_min"] <= rk["per_rank_images_per_sec_max"]
    # the line explains itself: every rank reports its own rate, the clock its GEMM launches got, its
    # device's socket power and its CPU binding; value = slowest-rank rate x ranks, next to the sum of the ranks' own rates
    assert [r["rank"] for r in rk["per_rank"]] == [0, 1]
    for r in rk["per_rank"]:
        for k in ("device", "images_per_sec", "sclk_in_kernel_mhz", "sclk_smi_mhz_mean", "socket_power_w_mean", "cpu_affinity"):
            assert k in r, k
        assert r["images_per_sec"] > 0 and (r["sclk_in_kernel_mhz"] is None or 500 < r["sclk_in_kernel_mhz"] < 2600)
    assert rk["images_per_sec_sum_of_rank_rates"] >= d["value"] * (1 - 1e-9)
    assert abs(rk["images_per_sec_sum_of_rank_rates"] - sum(r["images_per_sec"] for r in rk["per_rank"])) < 1e-6 * d["value"]
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"].startswith("dp2")
    # value = all ranks' images / the slowest rank's time
    assert abs(d["value"] - 2 * rk["per_rank_images_per_sec_min"]) / d["value"] < 1e-6


def test_two_rank_rehearsal_under_an_external_launcher():
    """The driver's own form: python -m torch.distributed.run ... bench.py --gpus 2 (the script is a rank)."""
    e = dict(os.environ, **REH)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    from robustvlm_amd.launch import free_port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), BENCH, "--gpus", "2", *SMALL],
                       capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)
    d = _one_line(r)
    assert d["n_gpus"] == 2 and d["ranks"]["launcher"] == "external torch.distributed.run"


def test_two_rank_rehearsal_dying_rank_exits_nonzero():
    r = run("--gpus", "2", *SMALL, env=dict(REH, RVLM_BENCH_DIE_RANK="1"), timeout=900)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no throughput line from a job that lost a rank"
    assert "a rank of the 2-GPU job failed" in r.stderr


def test_two_rank_rehearsal_train_step_line():
    """--mode train --gpus 2: the bucketed gradient all-reduce (through the host copy, gloo) inside bench.py's own loop."""
    d = _one_line(run("--gpus", "2", "--mode", "train", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "4",
                      "--iterations", "2", env=REH, timeout=900))
    assert d["n_gpus"] == 2 and "rehearsal" in d and d["value"] > 0
    assert d["allreduce"]["world"] == 2 and "gloo" in d["allreduce"]["backend"]
    assert d["final_loss"] == d["final_loss"]        # finite


def test_fp32_line_is_priced_against_the_fp32_matrix_peak():
    """--precision fp32 (the reference's own precision on v_mfma_f32_32x32x2_f32): dtype, roofline peak and kernel name of the line."""
    d = _one_line(run("--gpus", "1", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--precision", "fp32",
                      "--no-cpu-baseline"))
    assert d["dtype"] == "fp32" and d["roofline"]["peak"] == 157.3 and "gemm_f32_mfma_kernel" in d["roofline"]["kernel"]
    assert 0 < d["roofline"]["frac"] < 1


def test_mixed_precision_line():
    d = _one_line(run("--gpus", "1", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--precision",
                      "bf16+fp32-first", "--no-cpu-baseline"))
    assert d["dtype"] == "bf16+fp32-first" and d["value"] > 0 and d["roofline"]["peak"] == 2500.0


@pytest.mark.parametrize("attack", ["pgd", "apgd", "autopgd"])
def test_l2_norm_lines(attack):
    """scripts/bench_extra.py --norm l2: the three entry points under the L2 threat model through bench.py's harness; the line names
    it and is not labelled as a BASELINE config.  bench.py itself does not know the option (it only holds BASELINE's workloads)."""
    d = _one_line(run("--gpus", "1", "--steps", "1", "--warmup", "1", "--model", "ViT-B-32", "--batch", "8", "--attack", attack,
                      "--norm", "l2", "--no-cpu-baseline", script=BENCH_EXTRA))
    assert d["config"]["norm"] == "l2" and "L2" in d["metric"] and "BASELINE configs" not in d["config"]["workload"]
    assert d["value"] > 0


def test_headline_harness_refuses_the_extras():
    """bench.py only parses what BASELINE.json's configs need: --norm / --attack square / other models live in scripts/bench_extra.py."""
    for extra in (("--norm", "l2"), ("--attack", "square"), ("--model", "ViT-L-14-336")):
        r = run("--steps", "1", "--warmup", "0", *extra)
        assert r.returncode == 2 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (extra, r.stderr[-300:])
