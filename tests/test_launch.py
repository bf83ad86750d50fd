"""Host logic of the one-command multi-GPU launcher (robustvlm_amd/launch.py).  The reference gets
N GPUs from one command (train/adversarial_training_clip.py:184-191); `python bench.py --gpus N` must too."""
import os
import subprocess
import sys

import pytest

from robustvlm_amd.launch import LaunchError, plan_launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGV = ["bench.py", "--gpus", "4", "--steps", "2"]


def test_single_gpu_runs_in_process():
    p = plan_launch(1, ARGV, {}, visible_devices=1)
    assert p.role == "single" and p.world == 1 and p.cmd == []


def test_spawn_plan_is_one_rank_per_gpu_on_loopback():
    p = plan_launch(4, ARGV, {"PATH": "/usr/bin"}, visible_devices=8, port=29555)
    assert p.role == "spawn" and p.world == 4
    assert p.cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in p.cmd and "--nproc-per-node=4" in p.cmd
    assert p.cmd[p.cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert p.cmd[p.cmd.index("--master-port") + 1] == "29555"
    assert p.cmd[-len(ARGV):] == ARGV                      # the script's own arguments travel unchanged
    assert p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and p.env["RVLM_SELF_LAUNCHED"] == "1"
    assert "WORLD_SIZE" not in p.env                       # the agent sets the rendezvous, not the parent


def test_spawn_picks_a_free_port():
    p = plan_launch(2, ARGV, {}, visible_devices=2)
    assert 1024 < int(p.cmd[p.cmd.index("--master-port") + 1]) < 65536


def test_more_gpus_than_devices_fails_loudly(capsys):
    with pytest.raises(LaunchError) as e:
        plan_launch(2, ARGV, {}, visible_devices=1)
    assert e.value.code != 0 and "only 1 GPU" in e.value.msg
    assert "only 1 GPU" in capsys.readouterr().err
    with pytest.raises(LaunchError):
        plan_launch(1, ARGV, {}, visible_devices=0)
    with pytest.raises(LaunchError):
        plan_launch(0, ARGV, {}, visible_devices=8)


def test_under_a_launcher_the_process_is_a_rank():
    env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}
    p = plan_launch(8, ARGV, env, visible_devices=8)
    assert p.role == "rank" and p.world == 8 and p.cmd == []


def test_world_size_mismatch_is_an_error_not_a_warning():
    with pytest.raises(LaunchError) as e:
        plan_launch(8, ARGV, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, visible_devices=8)
    assert "WORLD_SIZE is 1" in e.value.msg
    with pytest.raises(LaunchError):
        plan_launch(1, ARGV, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, visible_devices=8)
    with pytest.raises(LaunchError):                       # a rank without a device of its own
        plan_launch(2, ARGV, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, visible_devices=1)


def test_self_launch_runs_n_ranks_and_propagates_failure(tmp_path):
    """The spawn plan executed for real on CPU ranks: N processes rendezvous over gloo; a dying rank makes the
    command exit non-zero."""
    from robustvlm_amd.launch import run_plan
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.ones(1); dist.all_reduce(t)\n"
        "assert int(t.item()) == int(os.environ['WORLD_SIZE']) == 2\n"
        "if dist.get_rank() == 0: open(sys.argv[1], 'w').write(str(int(t.item())))\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "sys.exit(int(sys.argv[2]) if os.environ['RANK'] == '1' else 0)\n")
    out = tmp_path / "ok.txt"
    p = plan_launch(2, [str(script), str(out), "0"], dict(os.environ), visible_devices=2)
    assert run_plan(p) == 0 and out.read_text() == "2"
    p = plan_launch(2, [str(script), str(out), "7"], dict(os.environ), visible_devices=2)
    assert run_plan(p) != 0


def test_bench_refuses_without_a_gpu():
    if __import__("torch").cuda.is_available():
        pytest.skip("CPU-container check")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)


def test_rehearsal_lets_ranks_share_a_device():
    """RVLM_BENCH_REHEARSAL=gloo: N ranks on fewer GPUs, so that the world > 1 branch of bench.py can
    run on a 1-GPU box; without the switch the same request stays an error."""
    env = {"RVLM_BENCH_REHEARSAL": "gloo"}
    p = plan_launch(2, ARGV, env, visible_devices=1, port=29556)
    assert p.role == "spawn" and p.world == 2 and p.env["RVLM_BENCH_REHEARSAL"] == "gloo"
    r = plan_launch(2, ARGV, dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), visible_devices=1)
    assert r.role == "rank" and r.world == 2
    with pytest.raises(LaunchError):
        plan_launch(2, ARGV, env, visible_devices=0)            # a rehearsal still needs a GPU
    with pytest.raises(LaunchError):
        plan_launch(2, ARGV, {"RVLM_BENCH_REHEARSAL": "mpi"}, visible_devices=2)
    with pytest.raises(LaunchError):
        plan_launch(2, ARGV, {}, visible_devices=1)
