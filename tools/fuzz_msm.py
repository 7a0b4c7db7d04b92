#!/usr/bin/env python3
"""Randomised differential run of the fixed-base MSM forms against the CPU oracle (not part of the test-suite: minutes of oracle time): for every table
width, N random blobs of several kinds -- uniform canonical elements, the bench recipe, sparse blobs, elements that are small multiples / negatives of
x^2 (GLV halves with extreme digits), all-equal elements -- through blob_to_kzg_commitment_many and compute_blob_kzg_proof_many, byte-exact against the
oracle's -march=native build on a thread pool.  usage: fuzz_msm.py [N per width] [widths ...]"""
import os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import kzg_rust_amd as kz
from oracle.oracle import Oracle, build
from synth import random_blob

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X2 = 0xd201000000010000 ** 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
widths = [int(a) for a in sys.argv[2:]] or [12, 13, 15, 16]
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
build(native=True)
o = Oracle(native=True)
so = o.load_trusted_setup(g1, g2)
rnd = random.Random(0x4844)


def blob(kind, i):
    if kind == 0:
        return b"".join(rnd.randrange(R).to_bytes(32, "big") for _ in range(4096))
    if kind == 1:
        return random_blob(900000 + i)
    if kind == 2:                                       # sparse: a few non-zero elements
        v = [0] * 4096
        for _ in range(rnd.randrange(1, 20)):
            v[rnd.randrange(4096)] = rnd.randrange(R)
        return b"".join(x.to_bytes(32, "big") for x in v)
    if kind == 3:                                       # halves with extreme digits: k = a + b x^2 with a, b near 0, x^2 - 1, 2^127, all-ones windows
        pick = lambda: rnd.choice([0, 1, X2 - 1, X2 - 2, 1 << 127, (1 << 127) - 1, (1 << 120) - 1, 0xAC45A400FFFF << 80, rnd.randrange(X2), (1 << rnd.randrange(1, 127)) - 1])
        out = []
        for _ in range(4096):
            k = pick() + pick() * X2
            out.append((k if k < R else k % R).to_bytes(32, "big"))
        return b"".join(out)
    e = rnd.randrange(R).to_bytes(32, "big")            # all elements equal
    return e * 4096


blobs = [blob(i % 5, i) for i in range(N)]
t0 = time.time()
with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
    want_c = list(ex.map(lambda b: o.blob_to_kzg_commitment(b, so), blobs))
    want_p = list(ex.map(lambda bc: o.compute_blob_kzg_proof(bc[0], bc[1], so), zip(blobs, want_c)))
print(f"oracle: {N} commitments + {N} proofs in {time.time() - t0:.1f} s", flush=True)
B = [kz.Blob(b) for b in blobs]
for bits in widths + ["glv-off-12"]:
    os.environ["KZG355_MSM_BITS"] = str(bits).split("-")[-1]
    if str(bits).startswith("glv-off"):
        os.environ["KZG355_MSM_GLV"] = "off"
    s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    os.environ.pop("KZG355_MSM_GLV", None)
    got_c = [c.to_bytes() for c in kz.Kzg.blob_to_kzg_commitment_many(B, s)]
    got_p = [p.to_bytes() for p in kz.Kzg.compute_blob_kzg_proof_many(B, [kz.KzgCommitment(c) for c in want_c], s)]
    bad_c = sum(a != b for a, b in zip(got_c, want_c)); bad_p = sum(a != b for a, b in zip(got_p, want_p))
    print(f"table {bits}: shape {s.msm_shape()[:3]}, {N} commitments: {bad_c} mismatches; {N} proofs: {bad_p} mismatches", flush=True)
    s.free()
    assert bad_c == 0 and bad_p == 0
print("all forms byte-exact against the oracle")
