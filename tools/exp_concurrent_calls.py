#!/usr/bin/env python3
"""T threads x R reference-shaped calls (verify_blob_kzg_proof_batch, n = 64, host slices) on ONE handle: per-thread median / worst call time and
the aggregate rate.  usage: exp_concurrent_calls.py [T ...]"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import kzg_rust_amd as kz
from synth import random_blob
g = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
L = kz.kzg.lib()
n = 64
blobs = [random_blob(7000 + i) for i in range(n)]
B = [kz.Blob(b) for b in blobs]
cs = kz.Kzg.blob_to_kzg_commitment_many(B, s); ps = kz.Kzg.compute_blob_kzg_proof_many(B, cs, s)
fb, fc, fp = b"".join(blobs), b"".join(c.to_bytes() for c in cs), b"".join(p.to_bytes() for p in ps)
R = 30
for T in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    times = [[] for _ in range(T)]
    gate = threading.Barrier(T)
    def work(k):
        ok = C.c_bool()
        for _ in range(3):
            L.kzg355_verify_blob_kzg_proof_batch(C.byref(ok), fb, n, fc, n, fp, n, s.handle)
        gate.wait()
        for r in range(R):
            t0 = time.perf_counter()
            rc = L.kzg355_verify_blob_kzg_proof_batch(C.byref(ok), fb, n, fc, n, fp, n, s.handle)
            times[k].append((time.perf_counter() - t0) * 1e3)
            assert rc == 0 and ok.value
    before = s.host_hashed_calls
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    med = [sorted(t)[len(t) // 2] for t in times]
    allt = sorted(x for t in times for x in t)
    print(f"T = {T}: median call per thread {[round(m, 2) for m in med]} ms, p90 {allt[int(0.9 * len(allt))]:.2f}, worst {allt[-1]:.2f} ms, host-hashed {s.host_hashed_calls - before} of {T * (R + 3)}, "
          f"aggregate {T * R * n / (sum(sum(t) for t in times) / T / 1e3):.0f} blobs/s", flush=True)
s.free()
