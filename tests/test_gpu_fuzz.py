"""-m gpu: the differential fuzzers of tools/ (fuzz_verify.py, fuzz_msm.py) with a fixed seed and a reduced count, so that the driver's own
`pytest -m gpu` run repeats them (the 30,000-batch / 12,000-blob runs of round 4 exist only as builder-written text under
profiles/r04/).  Pass rule as the reference's vector loops (src/lib.rs:189-201): Ok(true) / Ok(false) / Err of every batch equals the
oracle's, commitments and proofs are byte-exact.

  * 400 mutated batches of 1..12 blobs x 4 routes (one host-buffer call per batch, *_many on host buffers, device-resident submit / collect, device-resident synchronous calls),
    once per dispatch form: the defaults (few batches: pre-shifted / windowed lincomb, two-wave segmented pairing), then the forms only LARGE
    launch sets take by themselves, forced through the KZG355_* test overrides -- the bucket lincomb ending in one Horner chain per class,
    the final exponentiation's hard part twelve lanes per check, three Miller segments per pair;
  * 300 blobs of five kinds (uniform, bench recipe, sparse, GLV halves with extreme digits, all-equal) x the five MSM table forms."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

N_BATCHES = 400
N_BLOBS = 300

FORMS = {
    "defaults": {},
    "bucket lincomb with one chain per class + hard part twelve lanes per check": {"KZG355_LINCOMB": "bucket", "KZG355_LC_CHAIN_FROM": "1", "KZG355_PAIRING_HARD12_FROM": "1",
                                                                                    "KZG355_RHASH_LANES_FROM": "1"},
    "three Miller segments per pair + windowed lincomb + device hashes": {"KZG355_MILLER_SEGMENTS": "3", "KZG355_LINCOMB": "window", "KZG355_HOST_HASH": "off", "KZG355_HOST_RHASH": "off"},
}


@pytest.fixture(scope="module")
def verify_cases():
    import kzg_rust_amd as kz
    import fuzz_verify as fv
    s = fv.load_product(kz)
    blobs, cs, ps = fv.honest_pool(kz, s)
    s.free()
    cases = fv.make_cases(N_BATCHES, blobs, cs, ps, seed=0x4844_0005)
    want = fv.oracle_verdicts(cases)
    # the mutations must exercise all three outcomes, or the run proves nothing
    assert want.count(True) > 40 and want.count(False) > 100 and want.count(None) > 60, (want.count(True), want.count(False), want.count(None))
    return cases, want


@pytest.mark.parametrize("form", list(FORMS))
def test_verify_fuzz_three_routes(verify_cases, form):
    import kzg_rust_amd as kz
    import fuzz_verify as fv
    cases, want = verify_cases
    env = FORMS[form]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = fv.load_product(kz)             # (kzg355_load_trusted_setup reads the KZG355_* test overrides)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        bad = fv.run_routes(kz, s, cases, want)
    finally:
        s.free()
    assert bad == (0, 0, 0, 0), f"{form}: mismatches per route (single calls, *_many, submit / collect, synchronous device-resident) {bad} of {len(cases)} batches"


@pytest.fixture(scope="module")
def msm_cases():
    import fuzz_msm as fm
    blobs = fm.make_blobs(N_BLOBS, seed=0x4845)
    want_c, want_p = fm.oracle_outputs(blobs)
    return blobs, want_c, want_p


@pytest.mark.parametrize("form", [12, 13, 15, 16, "glv-off-12"])
def test_msm_fuzz_every_table_form(msm_cases, form):
    import kzg_rust_amd as kz
    import fuzz_msm as fm
    blobs, want_c, want_p = msm_cases
    bad_c, bad_p, shape = fm.run_form(kz, form, blobs, want_c, want_p)
    assert shape[0] == int(str(form).split("-")[-1]) and shape[2] == (0 if str(form).startswith("glv-off") else 1), shape
    assert (bad_c, bad_p) == (0, 0), f"table form {form} {shape}: {bad_c} commitments and {bad_p} proofs of {len(blobs)} differ from the oracle"


@pytest.mark.parametrize("form", [2, 4, 6])
def test_quotient_tree_every_lane_shape(msm_cases, form, oracle, oracle_settings, setup_bytes):
    """k_quotient_tree<LG> (quot_core.h): 4 / 16 / 64 leaves per lane.  The library picks 2^2 below 512 blobs and 2^4 above; KZG355_QUOTIENT_FORM pins one.
    Proofs of 60 blobs of the five kinds byte-exact against the oracle under each, and compute_kzg_proof at z outside and INSIDE the domain (the
    latter goes to the scan kernel through k_quotient_prep's list whatever the form; reference src/kzg.rs:461-528)."""
    import kzg_rust_amd as kz
    blobs, want_c, want_p = msm_cases
    g1, g2 = setup_bytes
    os.environ["KZG355_QUOTIENT_FORM"] = str(form)
    try:
        s = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)], msm_bits=12)
    finally:
        del os.environ["KZG355_QUOTIENT_FORM"]
    try:
        n = 60
        B = [kz.Blob(b) for b in blobs[:n]]
        got = kz.Kzg.compute_blob_kzg_proof_many(B, [kz.KzgCommitment(c) for c in want_c[:n]], s)
        assert [p.to_bytes() for p in got] == want_p[:n]
        R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
        w = pow(7, (R_ - 1) // 4096, R_)
        for zi in (5, R_ - 2, pow(w, 1, R_), pow(w, 2049, R_), 1, R_ - 1):       # the last four are roots of unity: z inside the domain
            z = zi.to_bytes(32, "big")
            for b in (blobs[1], blobs[3]):
                pr, y = kz.Kzg.compute_kzg_proof(kz.Blob(b), kz.Bytes32(z), s)
                wp, wy = oracle.compute_kzg_proof(b, z, oracle_settings)
                assert (pr.to_bytes(), y.to_bytes()) == (wp, wy), (form, hex(zi))
    finally:
        s.free()


def test_single_proof_many_forms_fuzz():
    """tools/fuzz_single_many.py with fixed seeds: 1500 mutated (C, z, y, proof) tuples through kzg355_verify_kzg_proof_many in one call, in calls of 40 and on
    device-resident records, and 96 (blob, z) pairs -- random, inside the domain, non-canonical -- through kzg355_compute_kzg_proof_many: every unit's
    Ok(true) / Ok(false) / Err and every proof / y byte agree with the oracle."""
    import kzg_rust_amd as kz
    import fuzz_single_many as fs
    import fuzz_verify as fv
    s = fv.load_product(kz)
    o, so = fs.make_oracle()
    try:
        blobs, tuples = fs.honest_tuples(o, so, n_blobs=4, per_blob=16)
        cases = fs.make_verify_cases(1500, tuples, seed=0x4844_0008)
        want = fs.oracle_verify(o, so, cases)
        assert want.count(True) > 150 and want.count(False) > 400 and want.count(None) > 300, (want.count(True), want.count(False), want.count(None))
        assert fs.run_verify_routes(kz, s, cases, want) == (0, 0, 0)
        bad, n_err = fs.run_compute(kz, s, o, so, blobs, 96, seed=0x4844_0009)
        assert bad == 0 and 2 <= n_err <= 30, (bad, n_err)
    finally:
        o.free_trusted_setup(so)
        s.free()
