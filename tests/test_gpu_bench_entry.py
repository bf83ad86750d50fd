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


def run(*a, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *a], capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


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
