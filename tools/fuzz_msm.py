#!/usr/bin/env python3
"""Randomised differential run of the fixed-base MSM forms against the CPU oracle: for every table width, N random blobs of several kinds -- uniform
canonical elements, the bench recipe, sparse blobs, elements that are small multiples / negatives of x^2 (GLV halves with extreme digits), all-equal
elements -- through blob_to_kzg_commitment_many and compute_blob_kzg_proof_many, byte-exact against the oracle's -march=native build on a thread pool.
As a tool: minutes of oracle time (profiles/r04/msm_fuzz.txt: 12,000 blobs x 5 forms); tests/test_gpu_fuzz.py runs the same functions with a fixed seed
and a few hundred blobs inside `pytest -m gpu`.
usage: fuzz_msm.py [N per width] [widths ...]"""
import os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X2 = 0xd201000000010000 ** 2
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
FORMS = (12, 13, 15, 16, "glv-off-12")


def setup_bytes():
    return open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), open(os.path.join(GOLDEN, "trusted_setup_g2.bin"), "rb").read()


def make_blobs(n, seed=0x4844):
    from synth import random_blob
    rnd = random.Random(seed)

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
    return [blob(i % 5, i) for i in range(n)]


def oracle_outputs(blobs, workers=None):
    from oracle.oracle import Oracle, build
    try:
        build(native=True); o = Oracle(native=True)
    except Exception:
        o = Oracle(native=False)
    so = o.load_trusted_setup(*setup_bytes())
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=workers or min(32, avail)) as ex:
        want_c = list(ex.map(lambda b: o.blob_to_kzg_commitment(b, so), blobs))
        want_p = list(ex.map(lambda bc: o.compute_blob_kzg_proof(bc[0], bc[1], so), zip(blobs, want_c)))
    o.free_trusted_setup(so)
    return want_c, want_p


def run_form(kz, form, blobs, want_c, want_p):
    """(mismatching commitments, mismatching proofs, table shape) of one table form: 12 / 13 / 15 / 16-bit GLV windows, "glv-off-12" = round 3's 256-bit windows"""
    g1, g2 = setup_bytes()
    opts = {"msm_bits": int(str(form).split("-")[-1]), "msm_require_wide": 1}
    if str(form).startswith("glv-off"):
        opts["msm_glv"] = -1
    s = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)], **opts)
    B = [kz.Blob(b) for b in blobs]
    got_c = [c.to_bytes() for c in kz.Kzg.blob_to_kzg_commitment_many(B, s)]
    got_p = [p.to_bytes() for p in kz.Kzg.compute_blob_kzg_proof_many(B, [kz.KzgCommitment(c) for c in want_c], s)]
    shape = s.msm_shape()[:3]
    s.free()
    return sum(a != b for a, b in zip(got_c, want_c)), sum(a != b for a, b in zip(got_p, want_p)), shape


def main():
    import kzg_rust_amd as kz
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    forms = [int(a) for a in sys.argv[2:]] or list(FORMS)
    seed = int(os.environ.get("KZG355_FUZZ_SEED", "0x4844"), 0)                 # another seed = another run (profiles/r06/msm_fuzz_seed2.txt)
    blobs = make_blobs(N, seed)
    print(f"seed {seed:#x}", flush=True)
    t0 = time.time()
    want_c, want_p = oracle_outputs(blobs)
    print(f"oracle: {N} commitments + {N} proofs in {time.time() - t0:.1f} s", flush=True)
    for form in forms:
        bad_c, bad_p, shape = run_form(kz, form, blobs, want_c, want_p)
        print(f"table {form}: shape {shape}, {N} commitments: {bad_c} mismatches; {N} proofs: {bad_p} mismatches", flush=True)
        assert bad_c == 0 and bad_p == 0
    print("all forms byte-exact against the oracle")


if __name__ == "__main__":
    main()
