"""-m gpu: the DPP-quad G1 routines (kzg_rust_amd/csrc/g1_quad.h: a doubling / an addition walked by four lanes per point) against
the plain lane routines of g1.h, on the device: generic operands, an operand at infinity, P = Q, P = -Q, lazy operands, doubling chains.
The binary is built by __graft_entry__.build() (tests/native/quad_ops_test.hip); it found a miscompile of hipcc 7.2 (DPP broadcasts
folded into their consumers) that the end-to-end vectors only showed as wrong verdicts."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quad_routines_match_lane_routines():
    exe = os.path.join(ROOT, "tests", "native", "quad_ops_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 7 and "MISMATCH" not in r.stdout, r.stdout
