#!/usr/bin/env python3
"""Randomised differential run of verify_blob_kzg_proof_batch against the CPU oracle.  N mutated batches of 1..12 blobs drawn from a pool of honest
(blob, commitment, proof) triples -- honest, a field element pushed to r / r - 1 / 2^256 - 1, a blob byte flipped, proofs swapped, a commitment
replaced by another blob's / by infinity / by a random x (off-curve or outside G1), flag bits or a byte of a proof corrupted -- through THREE routes of
the product: one host-buffer call per batch, all batches of equal size in one *_many call, the same on device-resident inputs through the
submit / collect pair (device hash), and -- round 5 -- through the synchronous device-resident call in chunks of at most 512 blobs (the route that copies
the blobs back and hashes them on the host).  Ok(true) / Ok(false) / Err must agree with the oracle for every batch on every route (pass rule of src/lib.rs:189-201).
As a tool: minutes of oracle time at the default N (profiles/r04/verify_fuzz.txt: 30,000 batches); tests/test_gpu_fuzz.py runs the same functions
with a fixed seed and a few hundred batches inside `pytest -m gpu`, once per dispatch form.
usage: fuzz_verify.py [N]"""
import ctypes as C, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
POOL = 48
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def setup_bytes():
    return open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read(), open(os.path.join(GOLDEN, "trusted_setup_g2.bin"), "rb").read()


def load_product(kz):
    g1, g2 = setup_bytes()
    return kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])


def honest_pool(kz, s, pool=POOL, seed0=770000):
    from synth import random_blob
    blobs = [random_blob(seed0 + i) for i in range(pool)]
    B0 = [kz.Blob(b) for b in blobs]
    cs = [c.to_bytes() for c in kz.Kzg.blob_to_kzg_commitment_many(B0, s)]
    ps = [p.to_bytes() for p in kz.Kzg.compute_blob_kzg_proof_many(B0, [kz.KzgCommitment(c) for c in cs], s)]
    return blobs, cs, ps


def make_cases(n_cases, blobs, cs, ps, seed=0x4844_0004):
    rnd = random.Random(seed)
    pool = len(blobs)

    def case():
        n = rnd.randrange(1, 13)
        idx = rnd.sample(range(pool), n)
        B, Cm, Pr = [blobs[i] for i in idx], [cs[i] for i in idx], [ps[i] for i in idx]
        k, j = rnd.randrange(11), rnd.randrange(n)
        if k == 1:
            v = rnd.choice([R, R - 1, (1 << 256) - 1, R + 5])
            b = bytearray(B[j]); e = rnd.randrange(4096); b[32 * e:32 * e + 32] = v.to_bytes(32, "big"); B[j] = bytes(b)
        elif k == 2:
            b = bytearray(B[j]); b[rnd.randrange(131072) | 1] ^= 1 << rnd.randrange(8); B[j] = bytes(b)
        elif k == 3 and n > 1:
            a = (j + 1) % n; Pr[j], Pr[a] = Pr[a], Pr[j]
        elif k == 4:
            Cm[j] = cs[(idx[j] + 1) % pool]
        elif k == 5:
            Cm[j] = bytes([0xC0]) + bytes(47)
        elif k == 6:
            x = bytearray(rnd.randrange(1 << 381).to_bytes(48, "big")); x[0] = (x[0] & 0x1F) | 0x80 | (0x20 if rnd.random() < .5 else 0)
            if rnd.random() < .5: Cm[j] = bytes(x)
            else: Pr[j] = bytes(x)
        elif k == 7:
            x = bytearray(Pr[j]); x[0] ^= rnd.choice([0x80, 0x40, 0x20]); Pr[j] = bytes(x)
        elif k == 8:
            x = bytearray(Pr[j]); x[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8); Pr[j] = bytes(x)
        elif k == 9:
            Pr[j] = bytes([0xC0]) + bytes(47)
        return B, Cm, Pr
    return [case() for _ in range(n_cases)]


def oracle_verdicts(cases, workers=None):
    """True / False / None (Err) per batch from the CPU oracle, on a thread pool (ctypes releases the GIL)"""
    from oracle.oracle import Oracle, OracleError, build
    try:
        build(native=True); o = Oracle(native=True)
    except Exception:
        o = Oracle(native=False)
    so = o.load_trusted_setup(*setup_bytes())

    def verdict(c):
        try:
            return o.verify_blob_kzg_proof_batch(c[0], c[1], c[2], so)
        except OracleError:
            return None
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=workers or min(32, avail)) as ex:
        want = list(ex.map(verdict, cases))
    o.free_trusted_setup(so)
    return want


def run_routes(kz, s, cases, want):
    """mismatch counts of the four routes: (one host-buffer call per batch, *_many on host buffers, device-resident submit / collect, device-resident
    synchronous calls of at most 512 blobs: the host-hash route of kzg355_verify_blob_kzg_proof_batch_many_device)"""
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", s.device)
    bad1 = 0
    for c, w in zip(cases, want):
        try:
            got = kz.Kzg.verify_blob_kzg_proof_batch([kz.Blob(b) for b in c[0]], [kz.KzgCommitment(x) for x in c[1]], [kz.KzgProof(x) for x in c[2]], s)
        except kz.Error:
            got = None
        bad1 += got != w
    by_n = {}
    for i, c in enumerate(cases):
        by_n.setdefault(len(c[0]), []).append(i)
    bad2 = bad3 = bad4 = 0
    before = s.host_hashed_calls
    for n, ids in sorted(by_n.items()):
        G = len(ids)
        fb = b"".join(b for i in ids for b in cases[i][0]); fc = b"".join(x for i in ids for x in cases[i][1]); fp = b"".join(x for i in ids for x in cases[i][2])
        ok = (C.c_bool * G)(); st = (C.c_int * G)()
        L.kzg355_verify_blob_kzg_proof_batch_many(ok, st, fb, fc, fp, n, G, s.handle)
        for k, i in enumerate(ids):
            got = None if st[k] else bool(ok[k])
            bad2 += got != want[i]
        tb = torch.frombuffer(bytearray(fb), dtype=torch.uint8).to(dev); tc = torch.frombuffer(bytearray(fc), dtype=torch.uint8).to(dev); tp = torch.frombuffer(bytearray(fp), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize()
        tk = C.c_void_p()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle) == 0
        ok3 = (C.c_bool * G)(); st3 = (C.c_int * G)()
        L.kzg355_verify_collect(tk, ok3, st3)
        for k, i in enumerate(ids):
            got = None if st3[k] else bool(ok3[k])
            bad3 += got != want[i]
        per = max(1, 512 // n)                                   # batches per synchronous call: <= 512 blobs, so every one of them takes the host-hash route
        for g0 in range(0, G, per):
            g = min(per, G - g0)
            ok4 = (C.c_bool * g)(); st4 = (C.c_int * g)()
            L.kzg355_verify_blob_kzg_proof_batch_many_device(ok4, st4, tb.data_ptr() + 131072 * n * g0, tc.data_ptr() + 48 * n * g0, tp.data_ptr() + 48 * n * g0, n, g, s.handle)
            for k in range(g):
                got = None if st4[k] else bool(ok4[k])
                bad4 += got != want[ids[g0 + k]]
    run_routes.host_hashed_device_calls = s.host_hashed_calls - before
    return bad1, bad2, bad3, bad4


def main():
    import kzg_rust_amd as kz
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    s = load_product(kz)
    blobs, cs, ps = honest_pool(kz, s)
    seed = int(os.environ.get("KZG355_FUZZ_SEED", "0x48440004"), 0)             # another seed = another run (profiles/r06/verify_fuzz_seed2.txt)
    cases = make_cases(N, blobs, cs, ps, seed)
    print(f"seed {seed:#x}", flush=True)
    t0 = time.time()
    want = oracle_verdicts(cases)
    print(f"oracle: {N} batches in {time.time() - t0:.1f} s: {want.count(True)} true, {want.count(False)} false, {want.count(None)} Err", flush=True)
    bad1, bad2, bad3, bad4 = run_routes(kz, s, cases, want)
    print(f"route 1 (one call per batch, host buffers): {bad1} mismatches; route 2 (*_many, host buffers): {bad2} mismatches; "
          f"route 3 (device-resident, submit / collect): {bad3} mismatches; route 4 (device-resident, synchronous calls of <= 512 blobs: host-hash route): {bad4} mismatches", flush=True)
    assert bad1 == 0 and bad2 == 0 and bad3 == 0 and bad4 == 0
    print("every verdict of every route agrees with the oracle")


if __name__ == "__main__":
    main()
