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


def test_flatten_scalars_puts_every_headline_figure_where_the_driver_keeps_it():
    """The driver's record of the bench line keeps scalars under config / roofline / cpu_baseline and drops nested objects and extra top-level keys
    (VERDICT r4): every figure BASELINE.md quotes must be such a scalar."""
    import bench
    line = {"config": {}, "roofline": {"per_kernel": {"eval": {"avg_launch_ms": 31.4, "frac": 0.37}, "challenge": {"avg_launch_ms": 38.5, "frac": 0.22}},
                                       "alu": {"path_frac_of_nominal": 0.457, "path_valu_wave_insts_per_blob": 126339.7, "path_frac_of_mix_floor": 0.86,
                                               "per_kernel": {"eval": {"frac_of_nominal": 0.43, "frac_of_mix_floor": 0.82}, "rpowers": {"frac_of_nominal": 0.13}}}},
            "cpu_baseline": {"value": 719.0, "all_cores": {"value": 10559.0, "threads": 64}}}
    host = {"single_call_ms": 1.98, "single_call_ms_min": 1.96, "single_call_blobs_per_s": 32365.4, "single_call_ms_device_hash": 6.0, "stream_blobs_per_s": 404323.9,
            "stream_h2d_gbps": 53.0}
    mid = {"blobs_per_s": 4061774.0, "blobs_per_s_one_set_at_a_time": 3595418.5}
    power = {"sclk_mhz": {"median": 2301.0}, "socket_power_w": {"median": 1237.0}}
    leg = lambda r: {"blobs_per_s": r, "ms_per_launch": 153.1, "g1_sweep_hbm_frac": 0.007, "traffic_over_algorithmic": 16.3, "gather_frac": 0.63, "kernel_ms": {"quotient": 4.0}}
    msm = {"commit": leg(106983.2), "proof": leg(100857.2), "msm_bits": 16, "table_gb": 143.5, "blobs_per_launch": 16384}
    bench.flatten_scalars(line, host, mid, power, msm)
    cfg, roof, cpu = line["config"], line["roofline"], line["cpu_baseline"]
    scalar = lambda v: isinstance(v, (int, float, str)) and not isinstance(v, bool)
    for key in ("single_call_ms", "single_call_ms_device_hash", "host_stream_blobs_per_s", "mid_size_blobs_per_s", "sclk_mhz_median", "commit_blobs_per_s", "commit_msm_bits",
                "commit_g1_sweep_hbm_frac", "commit_traffic_over_algorithmic", "commit_gather_frac", "proof_blobs_per_s", "proof_g1_sweep_hbm_frac",
                "proof_traffic_over_algorithmic", "proof_gather_frac", "proof_quotient_ms", "commit_table_gb"):
        assert key in cfg and scalar(cfg[key]), key
    for key in ("eval_ms", "eval_frac", "challenge_ms", "alu_path_frac_of_nominal", "alu_path_frac_of_mix_floor", "alu_eval_frac_of_mix_floor", "alu_eval_frac_of_nominal",
                "alu_rpowers_frac_of_nominal"):
        assert key in roof and scalar(roof[key]), key
    assert cpu["all_cores_value"] == 10559.0 and cpu["all_cores_threads"] == 64
