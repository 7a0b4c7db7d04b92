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


def test_a_failing_rank_fails_the_launcher():
    r = run(["--gpus", "2"], KZG355_BENCH_ECHO_RC="7")
    assert r.returncode != 0


def test_single_gpu_run_is_not_relaunched():
    r = run(["--gpus", "1", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    assert line["world"] == 1 and line["master_addr"] is None     # no torch.distributed.run in between
