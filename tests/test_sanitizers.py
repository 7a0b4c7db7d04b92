"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (SURVEY section 5 "race detection / sanitizers").  CPU only: the
GPU pool offers no device sanitizer, so what CAN be instrumented is
  * the oracle (oracle/bls12_381.c + kzg_oracle.c, `make -C oracle asan`) running the reference's 208 golden vectors, and
  * the HOST build of the device math headers (tests/native/hd_probe.cpp over csrc/field.h, g1.h, pairing_coop.h, pairing_lanes.h, eval_core.h,
    quot_core.h, modinv.h, sha256.h) running tests/test_device_math_host.py -- the same source the GPU kernels are compiled from.
Both run in a child Python with the sanitizer runtime preloaded (an instrumented shared object cannot be loaded into an uninstrumented process
otherwise); `-fno-sanitize-recover=undefined` turns any undefined behaviour into an abort, and AddressSanitizer aborts on its first finding, so a
zero exit code of the child's pytest means no finding.  (tests/test_host_pool.py covers ThreadSanitizer for the handle's host threads.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_under_sanitizers(pytest_args, **env):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan.so next to this gcc")
    e = dict(os.environ, LD_PRELOAD=asan + (":" + ubsan if ubsan else ""), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
             UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONMALLOC="malloc", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + pytest_args, cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    return r.stdout


def test_oracle_passes_the_reference_vectors_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    out = _run_under_sanitizers(["tests/test_oracle_vectors.py", "-k", "test_oracle_matches_reference_vectors"], KZG355_ORACLE_ASAN="1")
    assert "6 passed" in out, out[-500:]


def test_host_build_of_the_device_math_passes_under_asan_and_ubsan():
    out = _run_under_sanitizers(["tests/test_device_math_host.py"], KZG355_HD_PROBE_ASAN="1")
    assert " passed" in out and "failed" not in out, out[-500:]
