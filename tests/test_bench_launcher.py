"""`python bench.py --gpus N` starts its own N ranks: the parent launches torch.distributed.run as a CHILD before it has
imported torch or touched a GPU, relays rank 0's single JSON line and the child's exit code.  CPU only: the ranks run bench.py's echo hook."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KZG355_BENCH_ECHO="1", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_launcher_command_is_the_drivers():
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]


def test_gpus_2_launches_two_ranks_and_forwards_arguments():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--exchange", "allgather"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                              # ONE JSON line on stdout, whatever the launcher and the ranks print besides
    line = json.loads(lines[0])
    assert line["world"] == 2 and line["gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["exchange"] == "allgather"
    assert line["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--exchange", "allgather"]
    assert line["master_addr"] == "127.0.0.1"


def test_default_exchange_is_both_and_the_new_flags_reach_the_ranks():
    """round 5: one invocation times both exchange forms behind the parity gate; the flags that turn parts off are forwarded like the rest"""
    r = run(["--gpus", "2", "--steps", "2", "--no-parity-gate", "--no-in-library-leg"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["exchange"] == "both" and line["world"] == 2
    assert line["argv"] == ["--gpus", "2", "--steps", "2", "--no-parity-gate", "--no-in-library-leg"]
    import bench
    src = open(bench.__file__).read()
    assert "def parity_gate(" in src and "def in_library_leg(" in src and "value_exchange" in src


def test_too_few_devices_is_one_clear_line_and_a_non_zero_exit():
    """`--gpus N` with fewer than N visible devices: every rank says so and exits 3 before any rendezvous (this container has no GPU at all)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KZG355_BENCH_ECHO")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_a_failing_rank_fails_the_launcher():
    r = run(["--gpus", "2"], KZG355_BENCH_ECHO_RC="7")
    assert r.returncode != 0


def test_single_gpu_run_is_not_relaunched():
    r = run(["--gpus", "1", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["world"] == 1 and line["master_addr"] is None     # no torch.distributed.run in between
