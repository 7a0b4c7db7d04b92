"""Pins the CPU oracle (oracle/kzg_oracle.c) to the reference's own 208 golden vectors
(tests/golden/vectors.json, re-encoded from /root/reference/tests/**/data.yaml by
tests/golden/make_fixtures.py).  Runs on CPU (-m "not gpu")."""
import pytest

from vector_harness import run_function

EXPECTED_COUNTS = {
    "blob_to_kzg_commitment": 10, "compute_kzg_proof": 46, "compute_blob_kzg_proof": 14,
    "verify_kzg_proof": 92, "verify_blob_kzg_proof": 24, "verify_blob_kzg_proof_batch": 22,
}


@pytest.mark.parametrize("fn", list(EXPECTED_COUNTS))
def test_oracle_matches_reference_vectors(fn, golden_vectors, golden_blobs, oracle, oracle_settings):
    n, failures = run_function(fn, golden_vectors, oracle, oracle_settings, golden_blobs)
    assert n == EXPECTED_COUNTS[fn]
    assert not failures, "\n".join(failures)
