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


def _bench_line_inputs():
    line = {"config": {"workload": "w", "batch_size": 64, "batches_per_step": 8192, "blobs_per_step": 524288, "field_elements_per_blob": 4096, "sets_in_flight": 1,
                       "inputs": "resident in HBM", "msm_form": 15, "step_ms": {"median": 1.0}, "latency_ms_single_batch": 2.0, "latency_ms_single_batch_min": 1.9,
                       "host_inputs": {}, "mid_size_sets": {}, "power": {}, "skipped_for_time": "nothing"},
            "roofline": {"bound": "hbm", "kernel": "challenge", "achieved": 1790.4, "peak": 8000.0, "unit": "GB/s", "frac": 0.2238, "traffic": 6.94e10,
                         "traffic_unit": "bytes per launch", "traffic_source": "profiles/x.json", "algorithmic_bytes_per_launch": 6.88e10, "avg_launch_ms": 38.4, "launches": 20,
                         "kernel_timing": "HIP events", "kernel_ms_share": {}, "path_bytes_per_blob": 262240, "path_frac_of_hbm_peak": 0.151, "measured_stream_copy_gbps": 5112.3,
                         "note": "n", "per_kernel": {f: {"avg_launch_ms": 1.0 + i, "frac": 0.1} for i, f in enumerate(("eval", "challenge", "validate_points", "lincomb", "pairing",
                                                                                                                 "lincomb_horner", "rpowers"))},
                         "alu": {"path_frac_of_nominal": 0.457, "path_valu_wave_insts_per_blob": 126339.7, "path_frac_of_mix_floor": 0.86,
                                 "per_kernel": {"eval": {"frac_of_nominal": 0.43, "frac_of_mix_floor": 0.82}, "rpowers": {"frac_of_nominal": 0.13}}}},
            "cpu_baseline": {"value": 719.0, "unit": "blobs/s", "cores": 1, "kind": "port", "sample": "s", "primitives": "p", "portable_c_value": 460.0, "host_cpus": 256,
                             "cpu_model": "m", "threads_matched_value": 9000.0, "threads_matched_threads": 16, "all_cores": {"value": 10559.0, "threads": 64}}}
    host = {"single_call_ms": 1.98, "single_call_ms_min": 1.96, "single_call_blobs_per_s": 32365.4, "single_call_ms_device_hash": 6.0, "stream_blobs_per_s": 404323.9,
            "stream_h2d_gbps": 53.0, "single_call_host_threads": 16}
    mid = {"blobs_per_s": 4061774.0, "blobs_per_s_one_set_at_a_time": 3595418.5}
    power = {"sclk_mhz": {"median": 2301.0}, "socket_power_w": {"median": 1237.0}}
    leg = lambda r: {"blobs_per_s": r, "ms_per_launch": 153.1, "g1_sweep_hbm_frac": 0.007, "traffic_over_algorithmic": 16.3, "gather_frac": 0.63, "kernel_ms": {"quotient": 4.0}}
    msm = {"commit": leg(106983.2), "proof": leg(100857.2), "msm_bits": 16, "table_gb": 143.5, "blobs_per_launch": 16384}
    coexist = {"commit_default_blobs_per_s": 95000.0, "commit_default_table_gb": 68.9, "mixed_verify_blobs_per_s": 3.6e6, "mixed_commit_blobs_per_s": 93000.0}
    return line, host, mid, power, msm, coexist


def test_flatten_scalars_puts_every_headline_figure_where_the_driver_keeps_it():
    """The driver's record of the bench line keeps the first twenty scalars of config / roofline / cpu_baseline, in order, and drops nested objects and extra
    top-level keys: north_star's own numbers must be those twenty, low-value scalars live under a nested `detail`."""
    import bench
    line, host, mid, power, msm, coexist = _bench_line_inputs()
    bench.flatten_scalars(line, host, mid, power, msm, coexist)
    cfg, roof, cpu = line["config"], line["roofline"], line["cpu_baseline"]
    scalar = lambda v: isinstance(v, (int, float, str)) and not isinstance(v, bool)
    first = lambda d, k: [key for key, v in d.items() if not isinstance(v, (dict, list))][:k]
    assert first(cfg, 21) == ["workload", "batch_size", "batches_per_step", "inputs", "single_call_ms", "single_call_ms_device_hash", "single_call_host_threads",
                              "host_stream_blobs_per_s", "latency_ms_single_batch", "commit_blobs_per_s", "commit_msm_bits", "commit_g1_sweep_hbm_frac",
                              "commit_traffic_over_algorithmic", "commit_gather_frac", "proof_blobs_per_s", "proof_g1_sweep_hbm_frac", "sclk_mhz_median",
                              "commit_default_blobs_per_s", "commit_default_table_gb", "mixed_verify_blobs_per_s", "mixed_commit_blobs_per_s"]
    assert all(scalar(cfg[k]) for k in first(cfg, 21))
    assert first(roof, 20) == ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_bytes_per_launch", "path_frac_of_hbm_peak",
                               "alu_path_frac_of_nominal", "alu_path_frac_of_mix_floor", "challenge_ms", "eval_ms", "validate_ms", "lincomb_ms", "pairing_ms",
                               "measured_stream_copy_gbps", "path_bytes_per_blob", "alu_path_wave_insts_per_blob"]
    assert first(cpu, 9) == ["value", "unit", "cores", "kind", "sample", "threads_matched_value", "threads_matched_threads", "all_cores_value", "all_cores_threads"]
    # nothing is lost: the low-value scalars moved under `detail`, every other figure is still a scalar somewhere behind the twenty
    assert set(cfg["detail"]) >= {"blobs_per_step", "field_elements_per_blob", "sets_in_flight", "msm_form", "latency_ms_single_batch_min", "single_call_ms_min"}
    assert set(roof["detail"]) == {"traffic_unit", "traffic_source", "kernel_timing", "note"}
    for key in ("mid_size_blobs_per_s", "proof_traffic_over_algorithmic", "proof_gather_frac", "proof_quotient_ms", "commit_table_gb", "skipped_for_time"):
        assert key in cfg and scalar(cfg[key]), key
    for key in ("eval_frac", "challenge_frac", "alu_eval_frac_of_mix_floor", "alu_eval_frac_of_nominal", "alu_rpowers_frac_of_nominal", "lincomb_horner_ms", "launches"):
        assert key in roof and scalar(roof[key]), key
    assert cpu["all_cores_value"] == 10559.0 and cpu["all_cores_threads"] == 64 and isinstance(cpu["all_cores"], dict)
    assert isinstance(cfg["msm_legs"], dict) and isinstance(cfg["coexistence"], dict) and isinstance(roof["per_kernel"], dict)


def test_flatten_scalars_order_of_a_multi_gpu_line():
    """N > 1: the exchange forms, the parity gate, the stage times and the in-library leg are what the twenty slots go to."""
    import bench
    cfg = {"workload": "w", "batch_size": 512, "batches_per_step": 8192, "blobs_per_step": 1, "inputs": "resident in HBM", "latency_ms_single_batch": 5.0,
           "value_exchange": "allgather_split", "parity_gate": "passed", "parity_gate_blobs": 512, "parity_gate_r": "00", "skipped_for_time": "in-library leg",
           "exchange": {"mode": "x"}}
    for m in ("allgather", "allgather_split", "alltoall"):
        cfg[f"{m}_blobs_per_s"] = 1.0; cfg[f"{m}_ms_per_step"] = 2.0
        for k in ("stage1_ms", "exchange_ms", "stage2_ms", "merge_ms"):
            cfg[f"{k}_{m}"] = 3.0
    cfg.update({"in_library_single_call_ms": 5.5, "in_library_blobs_per_s": 1e6, "in_library_exchange": "ncclAllGather (RCCL)", "in_library_note": "n"})
    line = {"config": cfg, "roofline": None, "cpu_baseline": None}
    bench.flatten_scalars(line, None, None, {"sclk_mhz": {"median": 2300.0}, "socket_power_w": {"median": 1200.0}}, None, None, multi=True)
    keys = [k for k, v in line["config"].items() if not isinstance(v, (dict, list))][:20]
    assert keys == ["workload", "batch_size", "batches_per_step", "inputs", "value_exchange", "allgather_blobs_per_s", "allgather_split_blobs_per_s", "alltoall_blobs_per_s",
                    "parity_gate", "stage1_ms_allgather", "exchange_ms_allgather", "stage2_ms_allgather", "merge_ms_allgather", "exchange_ms_alltoall",
                    "stage2_ms_allgather_split", "in_library_single_call_ms", "in_library_blobs_per_s", "in_library_exchange", "skipped_for_time", "sclk_mhz_median"]
    assert "parity_gate_r" in line["config"]["detail"]


def test_max_seconds_reaches_the_ranks_and_defaults_under_the_drivers_timeout():
    """--max-seconds: forwarded like every flag; its default leaves two minutes of the driver's 600 s for start-up and the line."""
    r = run(["--gpus", "2", "--steps", "2", "--max-seconds", "333"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["argv"] == ["--gpus", "2", "--steps", "2", "--max-seconds", "333"]
    import bench
    src = open(bench.__file__).read()
    assert 'ap.add_argument("--max-seconds", type=float, default=480.0' in src and "skipped_for_time" in src
    # what is dropped, in order: the in-library leg first, then exchange forms beyond the first
    assert src.index('skipped.append(f"exchange form {m}")') < src.index('skipped.append("in-library leg")')


def test_an_interrupted_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to `bench.py --gpus N` (a harness timeout) is forwarded to the torch.distributed.run child; the launcher returns only once the
    child is gone and reports a non-zero exit."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KZG355_BENCH_ECHO="1", KZG355_BENCH_ECHO_SLEEP="60")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8.0)                                                # torchrun and both ranks are up (the ranks sleep in the echo hook)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert kids, "the launcher has no child yet"
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode != 0
    for k in kids:                                                 # the torch.distributed.run child is gone when the launcher returns
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", k
