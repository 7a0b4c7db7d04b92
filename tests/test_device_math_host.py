"""The DEVICE math headers (kzg_rust_amd/csrc/*.h: 29-bit-limb Fp/Fr, tower, G1, precomputed-line
pairing, SHA-256) compiled for the HOST with g++ and compared with the CPU oracle / Python big ints.
Runs on the build box (-m "not gpu"); the same functions run on the GPU through the C-ABI tests."""
import ctypes as C
import hashlib
import os
import random
import subprocess

import pytest

from oracle.pyref import P, R

HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(HERE, "native")


@pytest.fixture(scope="module")
def hd():
    src = os.path.join(NATIVE, "hd_probe.cpp")
    so = os.path.join(NATIVE, "libhd_probe.so")
    deps = [src] + [os.path.join(HERE, "..", "kzg_rust_amd", "csrc", f) for f in
                    ("field.h", "tower.h", "g1.h", "pairing.h", "pairing_coop.h", "pairing_lanes.h", "modinv.h", "sha256.h", "consts_gen.h", "eval_core.h", "quot_core.h", "fp6inv_tables.inc")]
    flags = ["-O2"]
    if os.environ.get("KZG355_HD_PROBE_ASAN") == "1":      # tests/test_sanitizers.py: the same probe under AddressSanitizer + UndefinedBehaviorSanitizer
        so = os.path.join(NATIVE, "libhd_probe_asan.so")
        # (-O0 and no -g: the instrumented compile of these headers takes 44 s this way, 2 min 10 s at -O1 and ~10 min with debug info; the run is a minute either way)
        flags = ["-O0", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++"] + flags + ["-shared", "-fPIC", "-o", so, src], check=True)
    return C.CDLL(so)


def _fp(hd, op, a, b=0):
    out = C.create_string_buffer(48)
    rc = hd.hd_fp_op(op, out, a.to_bytes(48, "big"), b.to_bytes(48, "big"))
    return rc, int.from_bytes(out.raw, "big")


def _fr(hd, op, a, b=0):
    out = C.create_string_buffer(32)
    rc = hd.hd_fr_op(op, out, a.to_bytes(32, "big"), b.to_bytes(32, "big"))
    return rc, int.from_bytes(out.raw, "big")


def test_fp_arithmetic(hd):
    rnd = random.Random(11)
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 380, (1 << 29) - 1, 1 << 29, (1 << 377) - 1]
    vals += [rnd.randrange(P) for _ in range(60)]
    for a in vals:
        for b in rnd.sample(vals, 5) + [0, P - 1]:
            assert _fp(hd, 0, a, b)[1] == (a + b) % P
            assert _fp(hd, 1, a, b)[1] == (a - b) % P
            assert _fp(hd, 2, a, b)[1] == (a * b) % P
        assert _fp(hd, 5, a)[1] == (-a) % P
        assert _fp(hd, 9, a)[1] == (a * a) % P            # dedicated squaring (mont_sqr)
        assert _fp(hd, 6, a)[1] == (2 * a) % P
        assert C.create_string_buffer(1) is not None
    for a in vals:                                   # divstep inversion (modinv.h) against Python
        assert _fp(hd, 8, a)[1] == (pow(a, -1, P) if a else 0), hex(a)
    for a in vals[:20]:
        if a:
            assert _fp(hd, 3, a)[1] == pow(a, -1, P)
        rc, s = _fp(hd, 4, a * a % P)
        assert rc == 0 and s * s % P == a * a % P
    assert _fp(hd, 0, P, 0)[0] == 1  # >= p rejected by the byte decoder


def test_fr_arithmetic(hd):
    rnd = random.Random(12)
    vals = [0, 1, R - 1, R - 2, R, R + 1, (1 << 256) - 1, 1 << 255] + [rnd.randrange(1 << 256) for _ in range(60)]
    for a in vals:
        for b in rnd.sample(vals, 5):
            assert _fr(hd, 0, a, b)[1] == (a + b) % R
            assert _fr(hd, 1, a, b)[1] == (a - b) % R
            assert _fr(hd, 2, a, b)[1] == (a * b) % R
        out = C.create_string_buffer(32)
        hd.hd_fr_op(4, out, a.to_bytes(32, "big"), bytes(32))
        assert out.raw[0] == (1 if a < R else 0)
    for _ in range(200):                             # fused a*b + c*d with one reduction (k_eval's S <- S d + q P)
        a, b, c, d = (rnd.choice(vals) for _ in range(4))
        out = C.create_string_buffer(32)
        assert hd.hd_fr_mul2(out, *(v.to_bytes(32, "big") for v in (a, b, c, d))) == 0
        assert int.from_bytes(out.raw, "big") == (a * b + c * d) % R
    for a in vals:
        assert _fr(hd, 5, a)[1] == (pow(a % R, -1, R) if a % R else 0), hex(a)
        assert _fr(hd, 6, a)[1] == a * a % R, hex(a)              # the dedicated squaring (mont_sqr<9>)
    for a in vals[:12]:
        if a % R:
            assert _fr(hd, 3, a)[1] == pow(a, -1, R)


def test_sha256(hd):
    rnd = random.Random(13)
    for n in [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 1000, 131152]:
        msg = bytes(rnd.randrange(256) for _ in range(n))
        out = C.create_string_buffer(32)
        hd.hd_sha256(out, msg, C.c_uint64(n))
        assert out.raw == hashlib.sha256(msg).digest(), n


G1_GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
INF = bytes([0xC0]) + bytes(47)


def test_g1_validate_matches_oracle(hd, oracle, golden_vectors):
    pts = {INF, G1_GEN}
    for fn in ("verify_kzg_proof", "verify_blob_kzg_proof"):
        for c in golden_vectors[fn]:
            for k in ("commitment", "proof"):
                h = c["input"][k]
                try:
                    b = bytes.fromhex(h[2:])
                except ValueError:
                    continue
                if len(b) == 48:
                    pts.add(b)
    rnd = random.Random(14)
    for _ in range(40):  # random x: on-curve ones are (almost surely) outside the subgroup
        b = bytearray(rnd.randrange(P).to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if rnd.random() < 0.5 else 0)
        pts.add(bytes(b))
    pts |= {bytes(48), bytes([0x80]) + bytes(47), bytes([0xE0]) + bytes(47), bytes([0xC0]) + bytes(46) + b"\x01",
            bytes([0x9A]) + b"\xff" * 47}
    n_ok = n_sub = 0
    for b in sorted(pts):
        out = C.create_string_buffer(48)
        rc = hd.hd_g1_validate(out, b, 1)
        want = oracle.g1_validate(b)
        assert (rc == 0) == (want == 0), b.hex()
        if rc == 0:
            assert out.raw == b
            n_ok += 1
        rc2 = hd.hd_g1_validate(out, b, 0)
        assert (rc2 == 0) == (oracle.g1_uncompress_only(b) == 0), b.hex()
        n_sub += rc2 == 0 and rc != 0
    assert n_ok >= 10 and n_sub >= 5  # both accept and "on curve, wrong subgroup" branches exercised


def test_g1_mul_add(hd, oracle, setup_bytes):
    rnd = random.Random(15)
    g1 = setup_bytes[0]
    pts = [G1_GEN, INF] + [g1[48 * i:48 * i + 48] for i in (0, 1, 77, 4095)]
    scal = [0, 1, 2, R - 1, R, R + 5, (1 << 256) - 1] + [rnd.randrange(R) for _ in range(4)]
    for p in pts:
        for k in rnd.sample(scal, 4) + [0, 1]:
            for q in (None, rnd.choice(pts)):
                kb = k.to_bytes(32, "big")
                out = C.create_string_buffer(48)
                assert hd.hd_g1_mul_add(out, p, kb, q) == 0
                assert out.raw == oracle.g1_mul_add(p, kb, q)
    # GLV split (used by k_lincomb): scalars are < r there
    for p in pts:
        for k in [0, 1, 2, R - 1, (1 << 128) - 1, 1 << 128, 0xac45a4010001a4020000000100000000, 0xac45a4010001a4020000000100000000 - 1] + [rnd.randrange(R) for _ in range(4)]:
            kb = k.to_bytes(32, "big")
            out = C.create_string_buffer(48)
            assert hd.hd_glv_mul(out, p, kb) == 0
            assert out.raw == oracle.g1_mul_add(p, kb, None), hex(k)
    # signed 4-bit window multiplication by a 128-bit scalar (k_lincomb_terms)
    for p in pts:
        for k in [0, 1, 7, 8, 9, 15, 16, (1 << 128) - 1, 0x88888888888888888888888888888888, 0x77777777777777777777777777777777,
                  0x8000000000000000_0000000000000000] + [rnd.randrange(1 << 128) for _ in range(4)]:
            out = C.create_string_buffer(48)
            assert hd.hd_w4_mul(out, p, k.to_bytes(16, "big")) == 0
            assert out.raw == oracle.g1_mul_add(p, k.to_bytes(32, "big"), None), hex(k)
    # add-or-double corner cases of the Jacobian+Jacobian routine
    two_g = oracle.g1_mul_add(G1_GEN, (2).to_bytes(32, "big"))
    neg_g = oracle.g1_mul_add(G1_GEN, (R - 1).to_bytes(32, "big"))
    for p, q, want in ((G1_GEN, G1_GEN, two_g), (G1_GEN, neg_g, INF), (INF, G1_GEN, G1_GEN), (G1_GEN, INF, G1_GEN), (INF, INF, INF)):
        out = C.create_string_buffer(48)
        assert hd.hd_g1_add_jac(out, p, q) == 0 and out.raw == want


def test_glv_split_by_barrett_division_matches_the_restoring_division(hd):
    """g1.h glv_split_fast (the fixed-base MSM's split, ~150 instructions) against glv_split (bit-serial) and Python: k = a + b x^2,
    0 <= a < x^2, for random scalars below r and the values where a Barrett estimate is off by one."""
    rng = random.Random(355)
    x2 = 0xd201000000010000 ** 2
    edge = [0, 1, x2 - 1, x2, x2 + 1, 2 * x2 - 1, 2 * x2, R - 1, R - 2, (x2 - 1) * x2, (x2 - 1) * x2 - 1, (x2 - 2) * x2 + x2 - 1,
            (1 << 255) - 1, (1 << 254), (1 << 127), (1 << 127) - 1, (1 << 128) - 1, (1 << 128), 3 * x2 - 1, 3 * x2]
    edge += [m * x2 + d for m in (1, 2, 12345, x2 // 2, x2 - 2) for d in (-1, 0, 1)]
    out = C.create_string_buffer(64)
    for k in edge + [rng.randrange(R) for _ in range(20000)]:
        assert hd.hd_glv_splits(out, k.to_bytes(32, "big")) == 0
        a, b, af, bf = (int.from_bytes(out.raw[16 * i:16 * i + 16], "little") for i in range(4))
        assert (a, b) == (k % x2, k // x2), hex(k)
        assert (af, bf) == (a, b), hex(k)


def test_g2_decompress_and_pairing(hd, oracle, setup_bytes):
    g1, g2 = setup_bytes
    for i in range(65):
        assert hd.hd_g2_decompress(g2[96 * i:96 * i + 96]) == 0
    bad = bytearray(g2[:96]); bad[95] ^= 1
    assert (hd.hd_g2_decompress(bytes(bad)) == 0) == (oracle.g2_uncompress_check(bytes(bad)) == 0)
    assert hd.hd_g2_decompress(bytes(96)) != 0
    q0, q1 = g2[:96], g2[96:192]
    rnd = random.Random(16)
    a = rnd.randrange(R)
    aG = oracle.g1_mul_add(G1_GEN, a.to_bytes(32, "big"))
    ok = C.c_int()
    # e(aG, Q) == e(G, ... ) style checks against the oracle on assorted inputs, incl. infinity
    cases = [(aG, q0, aG, q0), (aG, q0, G1_GEN, q0), (aG, q1, G1_GEN, q0), (INF, q0, INF, q1), (INF, q0, G1_GEN, q1),
             (g1[:48], q1, g1[48:96], q0), (g1[48:96], q0, g1[:48], q1)]
    for p1, qa, p2, qb in cases:
        want = oracle.pairings_verify(p1, qa, p2, qb)
        assert hd.hd_pairings_verify(C.byref(ok), p1, qa, p2, qb) == 0
        assert bool(ok.value) == want
        assert hd.hd_pairings_verify_coop(C.byref(ok), p1, qa, p2, qb) == 0      # wave-cooperative variant
        assert bool(ok.value) == want
        rc = hd.hd_pairings_verify_lanes12(C.byref(ok), p1, qa, p2, qb)          # hard part of the final exponentiation twelve lanes per check (pairing_lanes.h):
        assert rc == 0, rc                                                       # every coefficient of its result equals the cooperative run's
        assert bool(ok.value) == want
    # bilinearity: e([a]G, [tau]G2) == e([a][tau]... ) cannot be formed without tau; use e(aG, Q) == e(G, Q)^a via
    # e([a]G, Q) == e([a]G, Q) (true) and e([a]G, Q) == e([a+1]G, Q) (false)
    a1G = oracle.g1_mul_add(G1_GEN, ((a + 1) % R).to_bytes(32, "big"))
    assert hd.hd_pairings_verify(C.byref(ok), aG, q1, a1G, q1) == 0 and ok.value == 0
    assert hd.hd_pairings_verify_lanes12(C.byref(ok), aG, q1, a1G, q1) == 0 and ok.value == 0


def test_eval_group_of_four_matches_oracle(hd, oracle, oracle_settings, golden_blobs):
    """eval_core.h (what k_eval runs per lane) against the oracle's evaluate_polynomial_in_evaluation_form (kzg.rs:346-389):
    random z, z = 0 / 1 / r-1, and z INSIDE the domain (first, second, last position and a middle one), where the reference
    takes its special-case branch and the group fold relies on the polynomial identity instead."""
    from synth import random_blob, random_field_element
    R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    w = pow(7, (R_ - 1) // 4096, R_)
    brp = lambda i: int(format(i, "012b")[::-1], 2)
    in_domain = [pow(w, brp(pos), R_) for pos in (0, 1, 2, 3, 4, 2049, 4095)]
    zs = [random_field_element(100 + i) for i in range(4)] + [v.to_bytes(32, "big") for v in [0, 1, R_ - 1] + in_domain]
    def valid(b):
        try:
            return len(b) == 131072 and bool(oracle.blob_to_kzg_commitment(b, oracle_settings))
        except Exception:
            return False
    top = (R_ - 1).to_bytes(32, "big")
    # the lazy bounds of the tree (eval_core.h header) at their worst: every value r - 1, and r - 1 / 0 patterns that maximise the differences
    ones = (((R_ >> 232) << 232) - 1).to_bytes(32, "big")          # every 29-bit limb below the top one at its maximum: the widest product columns
    extremes = [top * 4096, (top + bytes(32)) * 2048, (bytes(32) + top) * 2048, (top + top + bytes(64)) * 1024, (bytes(64) + top + top) * 1024,
                ones * 4096, (ones + bytes(32)) * 2048, (bytes(64) + ones + ones) * 1024]
    blobs = [random_blob(4242), next(b for b in golden_blobs if valid(b)), bytes(131072)] + extremes
    out = C.create_string_buffer(32)
    for blob in blobs:
        for z in zs:
            assert hd.hd_eval_poly(out, blob, z) == 0
            _, y = oracle.compute_kzg_proof(blob, z, oracle_settings)
            assert out.raw == y, z.hex()
    bad = bytearray(blobs[0]); bad[32 * 77:32 * 78] = (R_).to_bytes(32, "big")
    assert hd.hd_eval_poly(out, bytes(bad), zs[0]) == 1


def test_quotient_tree_matches_the_definition_and_the_oracle(hd, oracle, oracle_settings, golden_blobs):
    """quot_core.h (what k_quotient_tree runs per lane; reference src/kzg.rs:461-490): y against the oracle's compute_kzg_proof, and every q_i against
    its definition q_i (w_i - z) = p_i - y in Python integers, for the three lane shapes of the kernel, on random / extreme blobs and z = random, 0, 1,
    r - 2, 2, 5 (r - 1 = w^(N/2) is IN the domain); z inside the domain is reported (the device gives those blobs to the scan kernel)."""
    from synth import random_blob, random_field_element
    R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    w = pow(7, (R_ - 1) // 4096, R_)
    brp = lambda i: int(format(i, "012b")[::-1], 2)
    dom = [pow(w, brp(i), R_) for i in range(4096)]
    top = (R_ - 1).to_bytes(32, "big")
    ones = (((R_ >> 232) << 232) - 1).to_bytes(32, "big")          # every 29-bit limb below the top one at its maximum
    blobs = [random_blob(4343), bytes(131072), top * 4096, (top + bytes(32)) * 2048, ones * 4096, (bytes(64) + ones + top) * 1024]
    zs = [random_field_element(300), random_field_element(301)] + [v.to_bytes(32, "big") for v in (0, 1 + 1, R_ - 2, 5)]
    out_q, out_y = C.create_string_buffer(131072), C.create_string_buffer(32)
    for bi, blob in enumerate(blobs):
        p = [int.from_bytes(blob[32 * i:32 * i + 32], "big") for i in range(4096)]
        for zi, zb in enumerate(zs):
            lgs = (2, 4, 6) if (bi < 2 and zi < 2) else (4,)
            for lg in lgs:
                assert hd.hd_quotient(out_q, out_y, blob, zb, lg) == 0, (bi, zi, lg)
                _, y = oracle.compute_kzg_proof(blob, zb, oracle_settings)
                assert out_y.raw == y, (bi, zi, lg)
                yv, zv = int.from_bytes(y, "big"), int.from_bytes(zb, "big")
                q = out_q.raw
                for i in range(4096):
                    qi = int.from_bytes(q[32 * i:32 * i + 32], "big")
                    assert qi < R_ and (qi * (dom[i] - zv) - (p[i] - yv)) % R_ == 0, (bi, zi, lg, i)
    for pos in (0, 1, 2049, 4095):
        assert hd.hd_quotient(out_q, out_y, blobs[0], dom[pos].to_bytes(32, "big"), 4) == 2
    bad = bytearray(blobs[0]); bad[32 * 77:32 * 78] = (R_).to_bytes(32, "big")
    assert hd.hd_quotient(out_q, out_y, bytes(bad), zs[0], 4) == 1


def test_g1_xyzz_accumulator(hd, oracle, setup_bytes):
    """g1x_add_mixed (8M + 2S extended-Jacobian mixed addition): sums with repeats (P + P: the doubling branch), P - P
    (infinity), infinity operands and ordinary points agree with the Jacobian routines."""
    g1, _ = setup_bytes
    P = [g1[48 * i:48 * i + 48] for i in range(6)]
    inf = bytes([0xC0]) + bytes(47)
    def neg(p):
        return bytes([p[0] ^ 0x20]) + p[1:]
    def ref_sum(pts):                                    # chain of the probe's Jacobian + mixed additions
        acc = inf
        for q in pts:
            out = C.create_string_buffer(48)
            assert hd.hd_g1_mul_add(out, acc, (1).to_bytes(32, "big"), q) == 0
            acc = out.raw
        return acc
    cases = [[P[0]], [P[0], P[0]], [P[0], neg(P[0])], [inf, P[1]], [P[1], inf, P[2]], [P[0], P[0], P[0], neg(P[0])],
             [P[0], P[1], P[2], P[3], P[4], P[5]], [P[3], neg(P[3]), P[3]], [inf, inf], [P[2], P[4], neg(P[2]), neg(P[4])]]
    for pts in cases:
        out = C.create_string_buffer(48)
        assert hd.hd_g1x_sum(out, b"".join(pts), len(pts)) == 0
        assert out.raw == ref_sum(pts), [p.hex()[:8] for p in pts]
        assert hd.hd_g1x_sum_lazy(out, b"".join(pts), len(pts)) == 0          # lazy form: same sums, incl. the rare branches
        assert out.raw == ref_sum(pts), [p.hex()[:8] for p in pts]
    # a long lazy chain (bounds of the unreduced coordinates must hold for any length) against the canonical chain
    import random
    rnd = random.Random(5)
    allp = [g1[48 * i:48 * i + 48] for i in range(4096)]
    pts = [rnd.choice(allp) if rnd.random() < .9 else neg(rnd.choice(allp)) for _ in range(600)]
    a, b = C.create_string_buffer(48), C.create_string_buffer(48)
    assert hd.hd_g1x_sum(a, b"".join(pts), len(pts)) == 0 and hd.hd_g1x_sum_lazy(b, b"".join(pts), len(pts)) == 0
    assert a.raw == b.raw


def test_subgroup_check_on_small_order_and_random_curve_points(hd):
    """The lazy [x^2]P chain behind g1_in_subgroup must agree with [r]P == infinity (return code 99 flags a disagreement) on
    points that are ON the curve but OUTSIDE G1: random curve points (order divisible by cofactor primes) and points of small
    order 3, 11, 33 (the chain runs through infinity and P = +-Q additions there), plus their sums with G1 points."""
    import random
    from oracle import pyref as pr
    rnd = random.Random(381)
    H = 0x396c8c005555e1568c00aaab0000aaab                 # cofactor of E(Fp)
    def random_curve_point():
        while True:
            x = rnd.randrange(pr.P)
            y2 = (x * x * x + 4) % pr.P
            y = pow(y2, (pr.P + 1) // 4, pr.P)
            if y * y % pr.P == y2:
                return (x, y)
    n_out = 0
    gen = pr.g1_uncompress(bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"))
    for trial in range(12):
        T = random_curve_point()
        cands = [T]
        full = pr.g1_mul(T, pr.R)                          # kills the G1 component: order divides the cofactor
        for q in (3, 11, 33):
            S = pr.g1_mul(full, H // q) if full is not None else None
            if S is not None:
                cands += [S, pr.g1_add(S, gen), pr.g1_add(S, pr.g1_mul(gen, 12345))]
        for pt in cands:
            if pt is None:
                continue
            out = C.create_string_buffer(48)
            rc = hd.hd_g1_validate(out, pr.g1_compress(pt), 1)
            in_g1 = pr.g1_mul(pt, pr.R) is None
            assert rc == (0 if in_g1 else 3), (trial, rc, in_g1)
            n_out += not in_g1
    assert n_out >= 20


def test_weighted_bucket_sum_in_lazy_xyzz(hd, setup_bytes):
    """k_lc_wsum's arithmetic on the host: 16 bucket sums parked as raw lazy XYZZ accumulators, W = sum_b b B_b by running sums with
    the lazy XYZZ + XYZZ addition (g1x_add_lazy2).  Cases: ordinary points; empty buckets (infinity operands); equal bucket sums
    (acc = sum: the doubling branch); a bucket that cancels the running accumulator (acc passes through infinity)."""
    g1, _ = setup_bytes
    P = [g1[48 * i:48 * i + 48] for i in range(64)]
    inf = bytes([0xC0]) + bytes(47)
    def neg(p):
        return bytes([p[0] ^ 0x20]) + p[1:]
    def ref(pts):                                       # sum_b b (P_{2(b-1)} + P_{2(b-1)+1}) by the probe's canonical scalar multiplication
        acc = inf
        for b in range(16):
            for k in range(2):
                out = C.create_string_buffer(48)
                assert hd.hd_g1_mul_add(out, pts[2 * b + k], (b + 1).to_bytes(32, "big"), acc) == 0
                acc = out.raw
        return acc
    cases = []
    cases.append(P[:32])
    c = list(P[:32]); c[0] = c[1] = inf; c[30] = c[31] = inf; c[10] = inf; cases.append(c)                  # empty buckets 1, 16; a one-point bucket
    c = list(P[:32]); c[28], c[29] = c[30], c[31]; cases.append(c)                                            # B_15 = B_16
    c = list(P[:32]); c[28], c[29] = neg(c[30]), neg(c[31]); cases.append(c)                                  # B_15 = -B_16: acc = infinity after one step
    c = [inf] * 32; cases.append(c)                                                                           # everything empty
    c = [inf] * 32; c[30] = P[5]; cases.append(c)                                                             # only bucket 16
    c = list(P[32:64]); c[3] = neg(c[2]); cases.append(c)                                                     # a bucket whose two items cancel
    for pts in cases:
        out = C.create_string_buffer(48)
        assert hd.hd_weighted_bucket_sum(out, b"".join(pts)) == 0
        assert out.raw == ref(pts)


def test_pairing_with_projective_arguments_and_split_miller_loops(hd, oracle, setup_bytes):
    """The pairing check as the kernels run it: G1 arguments as (X Z, Y, Z^3) of a Jacobian point with z != 1 (lines scaled by Z^3:
    the final exponentiation must kill the factor), the two Miller loops run separately and multiplied.  Same verdicts as the
    lane-level reference pairing on true, false and infinity cases, for several z."""
    g1, g2 = setup_bytes
    G2GEN, TAU2 = g2[:96], g2[96:192]
    P0, P1 = g1[:48], g1[48:96]
    inf = bytes([0xC0]) + bytes(47)
    cases = [(P0, G2GEN, P0, G2GEN), (P0, TAU2, P0, TAU2), (P0, G2GEN, P1, G2GEN), (inf, G2GEN, inf, TAU2), (P0, TAU2, P1, G2GEN), (inf, G2GEN, P1, TAU2)]
    for (a, qa, b, qb) in cases:
        want = C.c_int(-1)
        assert hd.hd_pairings_verify(C.byref(want), a, qa, b, qb) == 0
        for z in (0, 1, 7):
            got = C.c_int(-1)
            assert hd.hd_pairings_verify_coop_proj(C.byref(got), a, qa, b, qb, z) == 0
            assert got.value == want.value, (z, a.hex()[:8], b.hex()[:8])
        for k in (1, 2, 3, 4):          # the Miller loops in k segments per pair (k_pairing_coop_split)
            got = C.c_int(-1)
            assert hd.hd_pairings_verify_coop_segments(C.byref(got), a, qa, b, qb, k) == 0
            assert got.value == want.value, (k, a.hex()[:8], b.hex()[:8])


def test_window_shifts_started_from_x_alone(hd, setup_bytes):
    """k_ps_shift walks its doubling chain on Y^2 = X^3 + 4 s^3 from (s x, s^2, 1) and lets the consumer multiply Z by y: same points
    as the chain from (x, y, 1), for both signs of y, the generator, setup points and the point at infinity."""
    import ctypes as C
    g1, _ = setup_bytes
    pts = [g1[48 * i:48 * i + 48] for i in (0, 1, 17, 4095)]
    pts.append(bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"))
    flipped = bytearray(pts[0]); flipped[0] ^= 0x20; pts.append(bytes(flipped))           # the other square root: -P
    pts.append(bytes([0xc0]) + bytes(47))
    for p in pts:
        for k in (0, 1, 5, 125):
            out, ref = C.create_string_buffer(48), C.create_string_buffer(48)
            assert hd.hd_shift_from_x(out, ref, p, k) == 0
            assert out.raw == ref.raw, (p.hex(), k)
