"""One command, N GPUs: the launcher behind ``python bench.py --gpus N``.

The reference gets its GPUs from one command (``torch.nn.DataParallel`` inside one process,
train/adversarial_training_clip.py:184-191).  Here the unit is one PROCESS per GPU, so the
one command has to create the ranks itself: when a script is started without a rendezvous in its environment
(no ``WORLD_SIZE``) and is asked for N > 1 GPUs, it re-executes itself under ``torch.distributed.run`` with N
local ranks on 127.0.0.1 and a free port, and returns that agent's exit code (non-zero as soon as any rank dies;
the agent tears the other ranks down).  When the rendezvous IS in the environment (the driver's own
``python -m torch.distributed.run ... bench.py --gpus N``) the script is a rank and nothing is launched.

REHEARSAL (``RVLM_BENCH_REHEARSAL=gloo``): an N > 1 job on a node with fewer than N GPUs - rank r uses device
r mod visible_devices and the process group is gloo.  It exists so that the world > 1 branch of bench.py (self-launch,
process-group init, barriers, the gather of per-rank times, rank 0's JSON line, a dying rank's exit code) runs on the
1-GPU boxes this project is developed on BEFORE an 8-GPU node executes it for the first time; its line is flagged
``rehearsal`` and is not a measurement.

Everything here is host logic and runs without a GPU (tests/test_launch.py)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from dataclasses import dataclass, field


class LaunchError(SystemExit):
    """Raised for a request that cannot be honoured; carries a non-zero exit code and prints its message."""

    def __init__(self, msg: str, code: int = 2):
        print(f"bench launch error: {msg}", file=sys.stderr, flush=True)
        super().__init__(code)
        self.msg = msg


@dataclass
class LaunchPlan:
    role: str                       # "single" (run in this process), "rank" (already under a launcher), "spawn"
    world: int
    cmd: list = field(default_factory=list)
    env: dict = field(default_factory=dict)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def plan_launch(gpus: int, argv: list, environ: dict, visible_devices: int, port: int | None = None) -> LaunchPlan:
    """Decide what ``script --gpus N`` has to do.

    * rendezvous already in the environment: this process is a rank; ``--gpus`` must equal ``WORLD_SIZE`` (a silent
      mismatch would label an N-GPU line with the throughput of another world size) and its ``LOCAL_RANK`` must name
      a visible device;
    * no rendezvous, N == 1: run here;
    * no rendezvous, N > 1: N must not exceed the visible devices; otherwise plan the self-launch.
    """
    if gpus < 1:
        raise LaunchError(f"--gpus {gpus}: need at least one GPU")
    rehearsal = rehearsal_backend(environ)
    if rehearsal and visible_devices >= 1:
        visible_devices = max(visible_devices, gpus)       # ranks share devices (rank r -> device r mod visible)
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            raise LaunchError(f"--gpus {gpus} but the launcher's WORLD_SIZE is {world}: start the script with "
                              f"--gpus {world}, or without a launcher (it creates its ranks itself)")
        local = int(environ.get("LOCAL_RANK", "0"))
        if local >= max(visible_devices, 0):
            raise LaunchError(f"LOCAL_RANK {local} but only {visible_devices} GPU(s) are visible")
        return LaunchPlan("rank", world)
    if visible_devices < gpus:
        raise LaunchError(f"--gpus {gpus} but only {visible_devices} GPU(s) are visible on this node "
                          f"(one process per GPU; ranks never share a device)")
    if gpus == 1:
        return LaunchPlan("single", 1)
    port = port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(argv)
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    env["RVLM_SELF_LAUNCHED"] = "1"
    return LaunchPlan("spawn", gpus, cmd, env)


def rehearsal_backend(environ) -> str:
    """'' (a real job: one GPU per rank, RCCL) or 'gloo' (RVLM_BENCH_REHEARSAL=gloo: ranks may share a GPU)."""
    v = environ.get("RVLM_BENCH_REHEARSAL", "")
    if v not in ("", "0", "gloo"):
        raise LaunchError(f"RVLM_BENCH_REHEARSAL={v!r}: the only rehearsal backend is 'gloo'")
    return "gloo" if v == "gloo" else ""


def run_plan(plan: LaunchPlan) -> int:
    """Execute a "spawn" plan; the exit code is the elastic agent's (non-zero if any rank failed)."""
    assert plan.role == "spawn"
    proc = subprocess.run(plan.cmd, env=plan.env)
    return int(proc.returncode)


def ensure_ranks(gpus: int, argv: list | None = None) -> LaunchPlan:
    """Call first thing in ``main``.  Returns the plan for "single" / "rank"; for "spawn" it launches the ranks,
    waits, and EXITS this process with their exit code (rank 0 of the children prints the JSON line)."""
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    plan = plan_launch(gpus, list(sys.argv if argv is None else argv), dict(os.environ), n_dev)
    if plan.role == "spawn":
        rc = run_plan(plan)
        if rc != 0:
            print(f"bench launch error: a rank of the {plan.world}-GPU job failed (exit code {rc})", file=sys.stderr)
        raise SystemExit(rc)
    return plan
