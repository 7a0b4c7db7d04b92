#!/usr/bin/env python3
"""Generates tests/golden/batch64.json and batch512.json.

SELF-GOLDEN: produced by the repository's CPU oracle (oracle/), not by the reference -- the reference's own vectors stop at
n = 6 and its only n = 64 signal is its bench.  The oracle itself is pinned by the reference's 208 vectors
(tests/test_oracle_vectors.py); these fixtures extend that anchor to the benchmark's batch sizes so the GPU stages can be
diffed one by one:  blobs are the seeded bench recipe (tests/synth.py: splitmix64, seed 0x48440000 + index, byte 0 of every
element zeroed), commitments / proofs are honest, and the dump holds every z_i, y_i, the batch challenge r and the two
pairing inputs (compressed) of verify_kzg_proof_batch (kzg.rs:579-627).

    python tests/golden/make_batch_fixtures.py        # ~3 min on 8 cores
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
FIRST_INDEX = {64: 640000, 512: 512000}


def _setup():
    from oracle.oracle import Oracle
    o = Oracle()
    g1 = open(os.path.join(HERE, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(HERE, "trusted_setup_g2.bin"), "rb").read()
    return o, o.load_trusted_setup(g1, g2)


def _commit_prove(index):
    from synth import random_blob
    global _O
    try:
        _O
    except NameError:
        _O = _setup()
    o, s = _O
    b = random_blob(index)
    c = o.blob_to_kzg_commitment(b, s)
    return c.hex(), o.compute_blob_kzg_proof(b, c, s).hex()


def main():
    from synth import random_blob
    o, s = _setup()
    for n, first in FIRST_INDEX.items():
        with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            cp = list(ex.map(_commit_prove, range(first, first + n), chunksize=4))
        blobs = [random_blob(first + i) for i in range(n)]
        cs = [bytes.fromhex(c) for c, _ in cp]; ps = [bytes.fromhex(p) for _, p in cp]
        inter = o.verify_batch_intermediates(blobs, cs, ps, s)
        assert inter["ok"] is True
        swapped = list(ps); swapped[n // 2], swapped[n // 2 + 1] = swapped[n // 2 + 1], swapped[n // 2]
        assert o.verify_blob_kzg_proof_batch(blobs, cs, swapped, s) is False
        json.dump({"provenance": "self-golden: CPU oracle of this repository (pinned by the reference's 208 vectors), see make_batch_fixtures.py",
                   "recipe": "tests/synth.py random_blob(first_index + i)", "n": n, "first_index": first,
                   "commitments": [c for c, _ in cp], "proofs": [p for _, p in cp],
                   "z": [z.hex() for z in inter["z"]], "y": [y.hex() for y in inter["y"]],
                   "r": inter["r"].hex(), "proof_lincomb": inter["proof_lincomb"].hex(), "rhs": inter["rhs"].hex(),
                   "expect": True, "swapped_pair": [n // 2, n // 2 + 1], "expect_swapped": False},
                  open(os.path.join(HERE, f"batch{n}.json"), "w"), indent=0)
        print(f"batch{n}.json written")


if __name__ == "__main__":
    main()
