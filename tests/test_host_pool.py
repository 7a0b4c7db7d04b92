"""The handle's host worker threads (kzg_rust_amd/csrc/host_pool.h: per-call hashing jobs running at once, the slice copy) under ThreadSanitizer:
six caller threads x 200 jobs on seven workers -- every index exactly once, no data race reported.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_pool_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "host_pool_test")
    src = os.path.join(ROOT, "tests", "native", "host_pool_test.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 problems" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
