#!/usr/bin/env python3
"""Randomised differential run of the *_many forms of the single-proof functions against the CPU oracle (round 6).

verify_kzg_proof: N mutated (commitment, z, y, proof) tuples drawn from honest ones -- honest; y off by one; z replaced by another tuple's; proof /
commitment swapped with a neighbour's; commitment or proof replaced by infinity, by a random x (off-curve or outside G1), with flag bits or a byte
corrupted; z or y pushed to r, r + 5, 2^256 - 1 -- through THREE routes of the product: ONE kzg355_verify_kzg_proof_many call over all N (the
throughput kernels: four ladder lanes per check, many-batch pairing), calls of 40 (fewer than 64 checks: the pre-shifted latency form), and
kzg355_verify_kzg_proof_many_device on device-resident records.  compute_kzg_proof: M (blob, z) pairs, z random / inside the domain / non-canonical,
through ONE kzg355_compute_kzg_proof_many call: proof and y byte-exact, Err for Err.  Pass rule of src/lib.rs:189-201 per unit.
usage: fuzz_single_many.py [N [M]]"""
import ctypes as C, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from fuzz_verify import R, load_product, setup_bytes

INF = bytes([0xC0]) + bytes(47)


def make_oracle():
    from oracle.oracle import Oracle, build
    try:
        build(native=True); o = Oracle(native=True)
    except Exception:
        o = Oracle(native=False)
    return o, o.load_trusted_setup(*setup_bytes())


def honest_tuples(o, so, n_blobs=6, per_blob=24, seed0=880000):
    """(blob index, C, z, y, proof) made by the ORACLE: the product only ever checks them"""
    from synth import random_blob, random_field_element
    blobs = [random_blob(seed0 + i) for i in range(n_blobs)]
    cs = [o.blob_to_kzg_commitment(b, so) for b in blobs]
    w = pow(7, (R - 1) // 4096, R)
    jobs = []
    for i in range(n_blobs):
        for k in range(per_blob):
            z = random_field_element(seed0 + 100 * i + k)
            if k % 8 == 7:
                z = pow(w, 1 + 37 * k + i, R).to_bytes(32, "big")          # inside the domain (kzg.rs:494-523)
            jobs.append((i, z))

    def one(job):
        i, z = job
        p, y = o.compute_kzg_proof(blobs[i], z, so)
        return (i, cs[i], z, y, p)
    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as ex:
        return blobs, list(ex.map(one, jobs))


def make_verify_cases(n, tuples, seed=0x4844_0006):
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        _, c, z, y, p = rnd.choice(tuples)
        k = rnd.randrange(14)
        if k == 1:
            y = ((int.from_bytes(y, "big") + rnd.choice([1, R - 1])) % R).to_bytes(32, "big")
        elif k == 2:
            z = rnd.choice(tuples)[2]
        elif k == 3:
            p = rnd.choice(tuples)[4]
        elif k == 4:
            c = rnd.choice(tuples)[1]
        elif k == 5:
            c = INF
        elif k == 6:
            p = INF
        elif k == 7:
            x = bytearray(rnd.randrange(1 << 381).to_bytes(48, "big")); x[0] = (x[0] & 0x1F) | 0x80 | (0x20 if rnd.random() < .5 else 0)
            if rnd.random() < .5: c = bytes(x)
            else: p = bytes(x)
        elif k == 8:
            x = bytearray(p); x[0] ^= rnd.choice([0x80, 0x40, 0x20]); p = bytes(x)
        elif k == 9:
            x = bytearray(c); x[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8); c = bytes(x)
        elif k == 10:
            z = rnd.choice([R, R + 5, (1 << 256) - 1]).to_bytes(32, "big")
        elif k == 11:
            y = rnd.choice([R, R + 5, (1 << 256) - 1]).to_bytes(32, "big")
        elif k == 12:
            c, p, y = INF, INF, bytes(32)                           # the zero polynomial: commitment = proof = infinity, y = 0: true (kzg.rs:299-301)
        out.append((c, z, y, p))
    return out


def oracle_verify(o, so, cases):
    from oracle.oracle import OracleError

    def verdict(t):
        try:
            return o.verify_kzg_proof(t[0], t[1], t[2], t[3], so)
        except OracleError:
            return None
    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as ex:
        return list(ex.map(verdict, cases))


def run_verify_routes(kz, s, cases, want):
    import torch
    L = kz.kzg.lib()
    n = len(cases)
    cs, zs, ys, ps = (b"".join(t[k] for t in cases) for k in range(4))
    ok = (C.c_bool * n)(); st = (C.c_int * n)()
    L.kzg355_verify_kzg_proof_many(ok, st, cs, zs, ys, ps, n, s.handle)
    bad1 = sum((None if st[i] else bool(ok[i])) != want[i] for i in range(n))
    bad2 = 0
    for lo in range(0, n, 40):
        m = min(40, n - lo)
        ok2 = (C.c_bool * m)(); st2 = (C.c_int * m)()
        L.kzg355_verify_kzg_proof_many(ok2, st2, cs[48 * lo:], zs[32 * lo:], ys[32 * lo:], ps[48 * lo:], m, s.handle)
        bad2 += sum((None if st2[i] else bool(ok2[i])) != want[lo + i] for i in range(m))
    rec = torch.frombuffer(bytearray(b"".join(t[0] + t[1] + t[2] + t[3] for t in cases)), dtype=torch.uint8).to(torch.device("cuda", s.device))
    torch.cuda.synchronize()
    ok3 = (C.c_bool * n)(); st3 = (C.c_int * n)()
    L.kzg355_verify_kzg_proof_many_device(ok3, st3, rec.data_ptr(), n, s.handle)
    bad3 = sum((None if st3[i] else bool(ok3[i])) != want[i] for i in range(n))
    return bad1, bad2, bad3


def run_compute(kz, s, o, so, blobs, m, seed=0x4844_0007):
    from oracle.oracle import OracleError
    rnd = random.Random(seed)
    w = pow(7, (R - 1) // 4096, R)
    jobs = []
    for _ in range(m):
        k = rnd.randrange(10)
        z = rnd.randrange(R) if k < 7 else pow(w, rnd.randrange(4096), R) if k < 9 else rnd.choice([R, R + 1, (1 << 256) - 1])
        jobs.append((rnd.randrange(len(blobs)), z.to_bytes(32, "big")))

    def one(j):
        try:
            return o.compute_kzg_proof(blobs[j[0]], j[1], so)
        except OracleError:
            return None
    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as ex:
        want = list(ex.map(one, jobs))
    got = kz.Kzg.compute_kzg_proof_many([blobs[j[0]] for j in jobs], [j[1] for j in jobs], s)
    bad = 0
    for g, w_ in zip(got, want):
        if isinstance(g, Exception):
            bad += w_ is not None
        else:
            bad += w_ is None or (bytes(g[0]), bytes(g[1])) != (w_[0], w_[1])
    return bad, sum(x is None for x in want)


def main():
    import kzg_rust_amd as kz
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    s = load_product(kz)
    o, so = make_oracle()
    t0 = time.time()
    blobs, tuples = honest_tuples(o, so)
    seed = int(os.environ.get("KZG355_FUZZ_SEED", "0x48440006"), 0)
    cases = make_verify_cases(n, tuples, seed)
    want = oracle_verify(o, so, cases)
    print(f"oracle: {len(tuples)} honest tuples, {n} mutated checks in {time.time() - t0:.1f} s: {want.count(True)} true, {want.count(False)} false, {want.count(None)} Err", flush=True)
    b1, b2, b3 = run_verify_routes(kz, s, cases, want)
    print(f"verify_kzg_proof_many: one call of {n}: {b1} mismatches; calls of 40: {b2} mismatches; device-resident records: {b3} mismatches", flush=True)
    bc, nerr = run_compute(kz, s, o, so, blobs, m, seed + 1)
    print(f"compute_kzg_proof_many: one call of {m} (blob, z) pairs, {nerr} of them Err by the oracle: {bc} mismatches", flush=True)
    assert b1 == 0 and b2 == 0 and b3 == 0 and bc == 0
    print("every unit of every route agrees with the oracle")


if __name__ == "__main__":
    main()
