#!/usr/bin/env python3
"""Fixtures for the minimal preset (FIELD_ELEMENTS_PER_BLOB = 4; BASELINE.json configs[0]).

The reference snapshot has no minimal vectors (SURVEY top note 1), so these are ORACLE-derived ("self-golden"), flagged as
such, and cross-checked here by the independent pure-Python oracle/pyref.py and by a monomial-form identity:
  * setup_g1_monomial_first4.bin: the first four MONOMIAL points [tau^k]G1 of the reference's own data file
    /root/reference/testing_trusted_setups.json ("setup_G1"; data, not source) -- read in the build container only;
  * minimal.json: the size-4 Lagrange setup derived from them (oracle/minimal_setup.py), blobs (4 x 32 bytes), commitments,
    proofs at points inside and outside the domain, blob proofs, a 6-blob batch with its stage-2 intermediates.
Checks made before anything is written: C oracle == pyref on every output; commitment == sum_k coeff_k [tau^k]G1 with the
coefficients interpolated from the blob's evaluations (the defining property of a Lagrange-form setup).
Run from the repo root:  python tests/golden/make_minimal_fixtures.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import minimal_setup, pyref            # noqa: E402
from oracle.oracle import Oracle, OracleError      # noqa: E402
from synth import splitmix64_bytes                 # noqa: E402

R = minimal_setup.R
N = 4


def main():
    ref = json.load(open("/root/reference/testing_trusted_setups.json"))
    mono = [bytes.fromhex(x[2:]) for x in ref["setup_G1"][:N]]
    open(os.path.join(HERE, "setup_g1_monomial_first4.bin"), "wb").write(b"".join(mono))
    assert b"".join(mono[:2]) == open(os.path.join(HERE, "setup_g1_monomial_first2.bin"), "rb").read()
    g2 = open(os.path.join(HERE, "trusted_setup_g2.bin"), "rb").read()
    assert [bytes.fromhex(x[2:]) for x in ref["setup_G2"]] == [g2[96 * i:96 * i + 96] for i in range(65)]

    o = Oracle(preset="minimal")
    assert o.field_elements_per_blob == N
    lag = minimal_setup.lagrange_from_monomial(o, mono)
    assert lag == minimal_setup.lagrange_from_monomial_pyref(mono), "C oracle and pyref disagree on the derived setup"
    so = o.load_trusted_setup(b"".join(lag), g2)
    pyref.set_preset(N)
    ps = pyref.Settings(lag, [g2[96 * i:96 * i + 96] for i in range(65)])
    roots_brp = ps.roots
    assert o.roots_of_unity(so) == b"".join(x.to_bytes(32, "big") for x in roots_brp)

    def fe(v):
        return (v % R).to_bytes(32, "big")
    stream = splitmix64_bytes(0x4844_0004, 32 * N * 8)
    blobs = [b"".join(b"\x00" + stream[32 * (N * b + i) + 1:32 * (N * b + i) + 32] for i in range(N)) for b in range(8)]   # bench recipe: top byte 0
    blobs += [bytes(32 * N), fe(R - 1) * N, fe(5) * N, fe(1) + fe(2) + fe(3) + fe(4)]
    commitments, blob_proofs = [], []
    w4 = pow(7, (R - 1) // N, R)
    for b in blobs:
        c = o.blob_to_kzg_commitment(b, so)
        assert c == pyref.blob_to_kzg_commitment(b, ps)
        # monomial identity: interpolate p over the (bit-reversed) domain, then C = sum coeff_k [tau^k]G1
        evals = [int.from_bytes(b[32 * i:32 * i + 32], "big") for i in range(N)]
        coeffs = []
        for k in range(N):       # inverse DFT over the points roots_brp[i]
            coeffs.append(sum(evals[i] * pow(roots_brp[i], -k, R) for i in range(N)) * pow(N, -1, R) % R)
        assert c == o.g1_lincomb(mono, [fe(x) for x in coeffs], fast=False), "not a Lagrange basis of the size-4 domain"
        commitments.append(c)
        p = o.compute_blob_kzg_proof(b, c, so)
        assert p == pyref.compute_blob_kzg_proof(b, c, ps)
        assert o.verify_blob_kzg_proof(b, c, p, so) is True and pyref.verify_blob_kzg_proof(b, c, p, ps) is True
        blob_proofs.append(p)
    proofs = []
    zs = [0, 1, 2, R - 1, w4, pow(w4, 3, R), int.from_bytes(splitmix64_bytes(77, 32), "big") % R]
    for bi in (0, 1, 9, 11):
        for z in zs:
            pr, y = o.compute_kzg_proof(blobs[bi], fe(z), so)
            assert (pr, y) == pyref.compute_kzg_proof(blobs[bi], fe(z), ps)
            assert o.verify_kzg_proof(commitments[bi], fe(z), y, pr, so) is True
            proofs.append({"blob": bi, "z": fe(z).hex(), "proof": pr.hex(), "y": y.hex()})
    nb = 6
    inter = o.verify_batch_intermediates(blobs[:nb], commitments[:nb], blob_proofs[:nb], so)
    assert inter["ok"] is True and pyref.verify_blob_kzg_proof_batch(blobs[:nb], commitments[:nb], blob_proofs[:nb], ps) is True
    swapped = list(blob_proofs[:nb]); swapped[1], swapped[2] = swapped[2], swapped[1]
    assert o.verify_blob_kzg_proof_batch(blobs[:nb], commitments[:nb], swapped, so) is False
    bad_blob = fe(1) + b"\xff" * 32 + fe(3) + fe(4)
    try:
        o.blob_to_kzg_commitment(bad_blob, so); raise SystemExit("non-canonical blob accepted")
    except OracleError:
        pass
    # a truncated mainnet Lagrange setup (what src/trusted_setup.rs:151 would produce) loads, but is NOT a basis of the size-4 domain
    g1_main = open(os.path.join(HERE, "trusted_setup_g1.bin"), "rb").read()
    st = o.load_trusted_setup(g1_main[:48 * N], g2)
    assert o.blob_to_kzg_commitment(blobs[0], st) != commitments[0]
    o.free_trusted_setup(st)
    out = {
        "provenance": "oracle-derived (self-golden): C oracle built with -DN_FE=4, cross-checked by oracle/pyref.py and by the monomial identity "
                      "C = sum coeff_k [tau^k]G1; monomial points from the reference's testing_trusted_setups.json (setup_G1[0..4])",
        "field_elements_per_blob": N,
        "setup_g1_lagrange": [x.hex() for x in lag],
        "roots_of_unity_brp": [fe(x).hex() for x in roots_brp],
        "blobs": [b.hex() for b in blobs], "commitments": [c.hex() for c in commitments], "blob_proofs": [p.hex() for p in blob_proofs],
        "compute_kzg_proof": proofs,
        "batch": {"n": nb, "z": [z.hex() for z in inter["z"]], "y": [y.hex() for y in inter["y"]], "r": inter["r"].hex(),
                  "proof_lincomb": inter["proof_lincomb"].hex(), "rhs": inter["rhs"].hex(), "expect": True, "swapped_pair": [1, 2], "expect_swapped": False},
        "invalid_blob": bad_blob.hex(),
    }
    json.dump(out, open(os.path.join(HERE, "minimal.json"), "w"), indent=1)
    o.free_trusted_setup(so)
    print(f"minimal.json: {len(blobs)} blobs, {len(proofs)} compute_kzg_proof cases, batch of {nb}")


if __name__ == "__main__":
    main()
