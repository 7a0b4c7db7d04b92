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


def test_batch64_fixture_consistent_with_oracle(oracle, oracle_settings):
    """tests/golden/batch64.json (self-golden, see make_batch_fixtures.py): the committed commitments / proofs / z / y / r and
    pairing inputs are what the oracle computes today from the seeded blobs -- guards the fixture against drift."""
    import json, os
    from synth import random_blob
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "batch64.json")))
    n, first = fx["n"], fx["first_index"]
    blobs = [random_blob(first + i) for i in range(n)]
    cs = [bytes.fromhex(c) for c in fx["commitments"]]; ps = [bytes.fromhex(p) for p in fx["proofs"]]
    for i in (0, 31, 63):
        assert oracle.blob_to_kzg_commitment(blobs[i], oracle_settings) == cs[i]
        assert oracle.compute_blob_kzg_proof(blobs[i], cs[i], oracle_settings) == ps[i]
    inter = oracle.verify_batch_intermediates(blobs, cs, ps, oracle_settings)
    assert inter["ok"] is True
    assert [z.hex() for z in inter["z"]] == fx["z"] and [y.hex() for y in inter["y"]] == fx["y"]
    assert inter["r"].hex() == fx["r"] and inter["proof_lincomb"].hex() == fx["proof_lincomb"] and inter["rhs"].hex() == fx["rhs"]
    a, b = fx["swapped_pair"]
    ps[a], ps[b] = ps[b], ps[a]
    assert oracle.verify_blob_kzg_proof_batch(blobs, cs, ps, oracle_settings) is False


def test_fast_primitive_form_of_the_port_is_pinned_by_the_same_vectors(golden_vectors, golden_blobs, setup_bytes):
    """The -march=native build of the oracle (bench.py's cpu_baseline) carries a second form of its hot primitives on CPUs with BMI2 + ADX (+ SHA):
    mulx / adcx / adox Montgomery products and SHA-extension hashing (oracle/bls12_381.c).  Whatever this host's build contains, it passes the
    reference's 208 vectors with the fast form switched on and again with it switched off."""
    from oracle.oracle import Oracle, build
    build(native=True)
    o = Oracle(native=True)
    s = o.load_trusted_setup(*setup_bytes)
    try:
        for on in ([True, False] if o.has_fast_primitives else [False]):
            o.set_fast_primitives(on)
            for fn, want in EXPECTED_COUNTS.items():
                n, failures = run_function(fn, golden_vectors, o, s, golden_blobs)
                assert n == want and not failures, (on, fn, failures[:3])
    finally:
        o.set_fast_primitives(True)
        o.free_trusted_setup(s)
